"""GPU: the train-mode BACKWARD on the native operators (backward half; morig_amd/train_backward.py,
csrc/train_bwd.hip) against torch.autograd on the CPU restatement of the reference's modules (oracle/nets.py) in float64.

Criterion for a gradient tensor g against its reference r: max|g - r| <= tol * max(|r|_inf, floor) -- gradients are compared at
the scale of the tensor they belong to. The BLOCKS (dense layer, edge MLP, the weight-gradient GEMM) are held to 2e-4. Whole
networks in training mode are ill-conditioned in fp32 (batch statistics over few rows, arg-max near-ties), so there the
reference is evaluated in float32 as well and serves as the yardstick: see _grad_report."""
import copy

import pytest
import torch

from helpers import PARITY_LOG
from morig_amd import models, native, synth
from morig_amd import train_backward as TB
from morig_amd.native import Mat
from oracle import nets

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(autouse=True)
def _grad_on():
    """conftest.py runs every test under torch.no_grad(); these need the graph. Module initialisation draws from the global
    generator: seeded, so that a run is reproducible."""
    torch.manual_seed(1234)
    with torch.enable_grad():
        yield


def _randomise(mod, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in mod.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.copy_(torch.rand(m.num_features, generator=g) + 0.5)
                m.weight[::3] *= -1.0                                   # negative gammas: the max picks the smallest input
                m.bias.copy_(torch.randn(m.num_features, generator=g) * 0.2)
    return mod


def _rel(got, ref, floor=1e-6):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    return float((got - ref).abs().max()) / max(float(ref.abs().max()), floor)


def _check(name, got, ref, tol):
    """|got - ref| at the scale of the tensor: 99 % of the entries within tol, none beyond 100 tol, same direction to 1e-4.
    (A gradient entry next to a ReLU kink or an arg-max near-tie is decided by the forward's last bit -- on any fp32
    implementation -- and then differs by its full size: isolated, never systematic.)"""
    g, r = got.detach().cpu().double().flatten(), ref.detach().cpu().double().flatten()
    scale = max(float(r.abs().max()), 1e-6)
    e = (g - r).abs() / scale
    PARITY_LOG.append((f"backward:{name}", float(e.max()) * scale, scale, float(e.max())))
    q99 = float(torch.quantile(e, 0.99)) if e.numel() > 100 else float(e.max())
    cos = float(torch.dot(g, r) / (g.norm() * r.norm() + 1e-300))
    assert q99 <= tol and float(e.max()) <= 100 * tol and cos >= 1.0 - 1e-4, (name, q99, float(e.max()), cos)


def _graph(n, e, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.stack([torch.randint(0, n, (e,), generator=g), torch.randint(0, n, (e,), generator=g)])


@pytest.mark.parametrize("bwd,tol", [("f32", 2e-6), ("bf16x3", 2e-5)])
@pytest.mark.parametrize("rows,N,K,spread", [(1000, 32, 7, 0), (5000, 130, 259, 0), (40000, 256, 64, 0), (333, 1, 1, 0), (70000, 512, 835, 0),
                                             (30000, 128, 256, 12), (3001, 200, 328, 0), (9000, 1024, 1800, 0)])
def test_gemm_tn(rows, N, K, spread, bwd, tol, monkeypatch):
    """C = A^T B over the rows (the weight gradient): the exact-float32 MFMA kernel (MORIG_TRAIN_BWD=f32) to 2e-6, the default bf16 x 3
    split kernel to 2e-5 of the result's scale -- also on operands whose columns span 24 orders of magnitude (`spread`: column c scaled
    by 10^(-spread .. +spread), the range gradients live in and an fp16 split could not hold), with odd shapes, a device-side live row
    count and bit-identical repeats."""
    monkeypatch.setenv("MORIG_TRAIN_BWD", bwd)
    ops = native.get_ops()
    g = torch.Generator().manual_seed(rows)
    A = torch.randn(rows, (N + 3) // 4 * 4, generator=g)
    B = torch.randn(rows, K + 5, generator=g)
    if spread:
        A[:, :N] *= 10.0 ** torch.linspace(-spread, spread, N)
        B[:, :K] *= 10.0 ** torch.linspace(spread, -spread, K)
    want = A[:, :N].double().t() @ B[:, :K].double()
    got = ops.gemm_tn(Mat.of(A.to(DEV), 0, N), Mat.of(B.to(DEV), 0, K))
    # every entry against the scale of ITS dot product, |a_n| |b_k| (with `spread` the result spans 48 orders of magnitude; a 1 x 1
    # product has nothing else to be measured against)
    ref_scale = A[:, :N].double().norm(dim=0)[:, None] * B[:, :K].double().norm(dim=0)[None, :]
    assert float(((got.cpu().double() - want).abs() / ref_scale).max()) <= tol
    if not spread:
        assert _rel(got, want) <= 10 * tol
    live = torch.tensor([rows // 3], dtype=torch.int32, device=DEV)        # device-side row count (E' of a CSR)
    got = ops.gemm_tn(Mat.of(A.to(DEV), 0, N), Mat.of(B.to(DEV), 0, K), rows_dev=live)
    want = A[: rows // 3, :N].double().t() @ B[: rows // 3, :K].double()
    ref_scale = A[: rows // 3, :N].double().norm(dim=0)[:, None] * B[: rows // 3, :K].double().norm(dim=0)[None, :]
    assert float(((got.cpu().double() - want).abs() / ref_scale).max()) <= tol
    again = ops.gemm_tn(Mat.of(A.to(DEV), 0, N), Mat.of(B.to(DEV), 0, K), rows_dev=live)
    assert torch.equal(got, again), "fixed summation order"


@pytest.mark.parametrize("rows,cols,ld,c0", [(300000, 16, 16, 0), (70000, 250, 252, 0), (1200000, 64, 64, 0), (777, 35, 37, 0),
                                             (262145, 130, 136, 2), (1, 5, 8, 0)])
def test_column_statistics_kernels(rows, cols, ld, c0):
    """the two-pass fp64 column reductions (col_stats, bn_backward_stats) at edge-buffer sizes: many slabs, the float4 and the
    scalar loader (ld or window not 16-byte aligned), a device-side live row count, bit-identical repeats"""
    ops = native.get_ops()
    g = torch.Generator().manual_seed(rows + cols)
    X = (torch.randn(rows, ld, generator=g) * 3.0 + 0.7)
    D = torch.randn(rows, ld, generator=g)
    Xd, Dd = X.to(DEV), D.to(DEV)
    for live in (rows, max(1, rows // 3)):
        rd = None if live == rows else torch.tensor([live], dtype=torch.int32, device=DEV)
        x64 = X[:live, c0:c0 + cols].double()
        mean, var, cnt = ops.col_stats(Mat.of(Xd, c0, cols), rows_dev=rd)
        assert float(cnt) == live
        assert _rel(mean, x64.mean(0)) <= 1e-6 and _rel(var, x64.var(0, unbiased=False)) <= 1e-6
        m2, v2, _ = ops.col_stats(Mat.of(Xd, c0, cols), rows_dev=rd)
        assert torch.equal(mean, m2) and torch.equal(var, v2)
        rstd = torch.rsqrt(var + 1e-5)
        d64 = D[:live, c0:c0 + cols].double()
        xh = ((X[:live, c0:c0 + cols] - mean.cpu()) * rstd.cpu()).double()
        sdz, sdzx = ops.bn_backward_stats(Mat.of(Dd, c0, cols), Mat.of(Xd, c0, cols), mean, rstd, rows_dev=rd)
        scale = float(d64.abs().sum(0).max())
        assert float((sdz.cpu().double() - d64.sum(0)).abs().max()) <= 1e-6 * scale
        assert float((sdzx.cpu().double() - (d64 * xh).sum(0)).abs().max()) <= 1e-5 * float((d64 * xh).abs().sum(0).max())
        only, none = ops.bn_backward_stats(Mat.of(Dd, c0, cols), rows_dev=rd)
        assert none is None and torch.equal(only, sdz)
        # BatchNorm + ReLU backward: the flat kernel, and the slab kernel that also leaves the column sums of du (the same du up to the
        # compiler's choice of fused multiply-adds)
        gam = torch.randn(cols, generator=g).to(DEV)
        du_a, du_b = torch.full((rows, ld), 2.0, device=DEV), torch.full((rows, ld), 2.0, device=DEV)
        assert ops.bn_relu_backward(Mat.of(Dd, c0, cols), Mat.of(Xd, c0, cols), mean, rstd, gam, sdz, sdzx, Mat.of(du_a, c0, cols), rows_dev=rd) is None
        sdu = ops.bn_relu_backward(Mat.of(Dd, c0, cols), Mat.of(Xd, c0, cols), mean, rstd, gam, sdz, sdzx, Mat.of(du_b, c0, cols), rows_dev=rd,
                                   want_sum=True)
        assert torch.allclose(du_a, du_b, rtol=2e-6, atol=1e-6 * float(du_a.abs().max()))
        assert torch.equal(du_a[live:], du_b[live:]) and torch.equal(du_a[:, :c0], du_b[:, :c0]) and torch.equal(du_a[:, c0 + cols:], du_b[:, c0 + cols:])
        ref = du_b[:live, c0:c0 + cols].double().sum(0)
        assert float((sdu.double() - ref).abs().max()) <= 1e-6 * float(du_b[:live, c0:c0 + cols].abs().double().sum(0).max())


@pytest.mark.parametrize("n,affine,track", [(1, True, True), (130, True, True), (1024, False, True), (77, True, False)])
def test_bn_finalize_is_functional_batch_norm(n, affine, track):
    """morig_bn_finalize against torch.nn.functional.batch_norm in training mode: the normalised output through (s, t), rstd, the
    running buffers (unbiased variance, momentum 0.1) and num_batches_tracked"""
    ops = native.get_ops()
    g = torch.Generator().manual_seed(n)
    rows = 57
    x = torch.randn(rows, n, generator=g) * 2.0 + 0.5
    bn = torch.nn.BatchNorm1d(n, affine=affine, track_running_stats=track)
    if affine:
        with torch.no_grad():
            bn.weight.copy_(torch.randn(n, generator=g)); bn.bias.copy_(torch.randn(n, generator=g))
    ref = copy.deepcopy(bn).train()
    want = ref(x)
    bn = bn.to(DEV)
    mean = x.mean(0).to(DEV)
    var = x.var(0, unbiased=False).to(DEV)
    cnt = torch.tensor([float(rows)], device=DEV)
    s, t, rstd = ops.bn_finalize(bn, mean, var, cnt)
    got = x.to(DEV) * s + t
    assert _rel(got, want) <= 1e-5
    assert _rel(rstd, torch.rsqrt(x.var(0, unbiased=False) + bn.eps)) <= 1e-6
    if track:
        assert _rel(bn.running_mean, ref.running_mean) <= 1e-6 and _rel(bn.running_var, ref.running_var) <= 1e-6
        assert int(bn.num_batches_tracked) == int(ref.num_batches_tracked) == 1


@pytest.mark.parametrize("rows,K,N", [(700, 35, 64), (4096, 832, 128), (257, 3, 16), (513, 20, 30)])
def test_dense_block_backward(rows, K, N):
    layer = _randomise(nets.mlp_stack([K, N]), K)[0].train()
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, K, generator=g)
    w = torch.randn(rows, N, generator=g)
    ref = copy.deepcopy(layer).double()
    xr = x.double().requires_grad_(True)
    want = ref(xr)
    (want * w.double()).sum().backward()
    mine = copy.deepcopy(layer).to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    out = TB.mlp_layer(xg, mine)
    (out * w.to(DEV)).sum().backward()
    _check("dense:out", out, want, 2e-5)
    _check("dense:dx", xg.grad, xr.grad, 2e-4)
    for (k, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
        _check(f"dense:{k}", p.grad, q.grad, 2e-4)
    assert torch.allclose(mine[2].running_mean.cpu().double(), ref[2].running_mean, atol=1e-5)


@pytest.mark.parametrize("n,e,C,H", [(300, 2000, 3, 16), (1500, 9000, 64, 128), (90, 400, 35, 32)])
def test_edge_mlp_backward(n, e, C, H):
    ops = native.get_ops()
    conv = _randomise(nets.EdgeMaxConv(C, H), n).train()
    ei = _graph(n, e, n)
    g = torch.Generator().manual_seed(e)
    x = torch.randn(n, C, generator=g)
    w = torch.randn(n, H, generator=g)
    ref = copy.deepcopy(conv).double()
    xr = x.double().requires_grad_(True)
    want = ref(xr, ei)
    (want * w.double()).sum().backward()
    mine = copy.deepcopy(conv).to(DEV)
    xg = x.to(DEV).requires_grad_(True)
    csr = ops.csr_build(ei.to(DEV), n)
    out = TB.edge_mlp(xg, csr, mine.nn_pos)
    (out * w.to(DEV)).sum().backward()
    _check("edge:out", out, want, 2e-5)
    _check("edge:dx", xg.grad, xr.grad, 2e-4)
    for (k, p), (_, q) in zip(mine.named_parameters(), ref.named_parameters()):
        _check(f"edge:{k}", p.grad, q.grad, 2e-4)


def _emu():
    from emulate import EmuOps
    return EmuOps()


@pytest.mark.parametrize("n_seg,rows_per,H,ld", [(8, 4096, 128, 128), (3, 1000, 35, 37), (5, 0, 16, 16), (2, 70000, 512, 512),
                                                 (40000, 9, 64, 64), (300, 7, 33, 36), (3000, 5, 35, 37)])
def test_segmax_arg_short_and_long_segments(n_seg, rows_per, H, ld):
    """few long segments (the pooling kernel: row lanes + LDS tree) and many short ones (one thread per segment and column group):
    the value, the FIRST winning row on ties, the winner's raw value, empty segments"""
    ops = native.get_ops()
    g = torch.Generator().manual_seed(n_seg * 31 + H)
    lens = torch.full((n_seg,), rows_per, dtype=torch.int64)
    if rows_per > 4:
        lens = lens + torch.randint(-3, 4, (n_seg,), generator=g)
    lens[n_seg // 2] = 0                                                   # an empty segment in the middle
    rowptr = torch.zeros(n_seg + 1, dtype=torch.int32)
    rowptr[1:] = lens.cumsum(0).int()
    rows = int(rowptr[-1])
    Z = torch.randn(max(rows, 1), ld, generator=g)
    Z[:, : H] = (Z[:, : H] * 4).round() / 4                                # many exact ties
    scale = torch.where(torch.arange(H) % 3 == 0, -1.0, 1.5) * (torch.rand(H, generator=g) + 0.5)
    shift = torch.randn(H, generator=g)
    emu = _emu()
    for sc, sh in ((scale, shift), (None, None)):
        want = torch.zeros(n_seg, H)
        warg, wz = emu.segmax_affine_arg(Mat.of(Z, 0, H), rowptr, n_seg, Mat.of(want, 0, H), sc, sh, want_zwin=True)
        out = torch.full((n_seg, ld), 7.0, device=DEV)
        arg, zwin = ops.segmax_affine_arg(Mat.of(Z.to(DEV), 0, H), rowptr.to(DEV), n_seg, Mat.of(out, 0, H),
                                          None if sc is None else sc.to(DEV), None if sh is None else sh.to(DEV), want_zwin=True)
        assert torch.equal(arg.cpu(), warg), (arg.cpu() != warg).nonzero()[:5]
        assert torch.equal(zwin.cpu(), wz)
        assert torch.allclose(out[:, :H].cpu(), want, rtol=1e-6, atol=1e-6)
        assert bool((out[:, H:] == 7.0).all())
        only = ops.segmax_affine_arg(Mat.of(Z.to(DEV), 0, H), rowptr.to(DEV), n_seg, Mat.of(out, 0, H),
                                     None if sc is None else sc.to(DEV), None if sh is None else sh.to(DEV))
        assert torch.equal(only.cpu(), warg)


@pytest.mark.parametrize("n,e,H,ld", [(500, 6000, 64, 64), (1200, 5000, 33, 36), (64, 3000, 16, 16), (3000, 40000, 256, 256)])
def test_edge_backward_operators(n, e, H, ld):
    """the three row passes of the edge MLP backward against the emulation: statistics from the kept winners (no gather), du2 with
    its column sums from one pass, BatchNorm1 + ReLU evaluated inside the two fixed-order scatter sums (bit-reproducible)"""
    ops, emu = native.get_ops(), _emu()
    g = torch.Generator().manual_seed(n + e)
    ei = _graph(n, e, n + 1)
    csr_d = ops.csr_build(ei.to(DEV), n)
    E = int(csr_d.rowptr[-1])
    from morig_amd.native import CSR
    csr = CSR(csr_d.rowptr.cpu(), csr_d.src.cpu(), csr_d.dst.cpu(), n, csr_d.capacity, csr_d.status.cpu())
    cap = csr.capacity
    Z = torch.randn(cap, ld, generator=g)
    Z[E:] = float("nan")                                                       # rows past the live count are never read
    mean, rstd = Z[:E, :H].mean(0), 1.0 / Z[:E, :H].std(0)
    gamma = torch.randn(H, generator=g)
    dout = torch.randn(n, ld, generator=g)
    dev = lambda t: t.to(DEV)
    out = torch.zeros(n, H)
    arg, zwin = emu.segmax_affine_arg(Mat.of(Z, 0, H), csr.rowptr, n, Mat.of(out, 0, H), gamma * rstd, -mean * gamma * rstd, want_zwin=True)
    # (1) statistics: kept winners vs gathered rows vs emulation
    want = emu.segmax_bn_backward_stats(Mat.of(dout, 0, H), arg, Mat.of(Z, 0, H), mean, rstd)
    Zd, doutd, argd = dev(Z), dev(dout), dev(arg)
    got_w = ops.segmax_bn_backward_stats(Mat.of(doutd, 0, H), argd, None, dev(mean), dev(rstd), zwin=dev(zwin))
    got_g = ops.segmax_bn_backward_stats(Mat.of(doutd, 0, H), argd, Mat.of(Zd, 0, H), dev(mean), dev(rstd))
    for a, b, w in zip(got_w, got_g, want):
        scale = float(w.abs().max()) + 1e-6
        assert float((a.cpu() - w).abs().max()) <= 2e-6 * scale * (n ** 0.5) and float((b.cpu() - w).abs().max()) <= 2e-6 * scale * (n ** 0.5)
    sdz, sdzx = want
    # (2) du2 + its column sums
    du_w = torch.full((cap, ld), 3.0)
    sum_w = emu.segmax_bn_relu_backward(Mat.of(dout, 0, H), arg, Mat.of(Z.nan_to_num(0.0), 0, H), csr.rowptr, csr.dst, mean, rstd, gamma, sdz,
                                        sdzx, Mat.of(du_w, 0, H), want_sum=True)
    du = torch.full((cap, ld), 3.0, device=DEV)
    sum_g = ops.segmax_bn_relu_backward(Mat.of(doutd, 0, H), argd, Mat.of(Zd, 0, H), dev(csr.rowptr), dev(csr.dst), dev(mean), dev(rstd),
                                        dev(gamma), dev(sdz), dev(sdzx), Mat.of(du, 0, H), want_sum=True)
    assert torch.allclose(du.cpu(), du_w, rtol=1e-5, atol=1e-6 * float(du_w.abs().max()))
    assert float((sum_g.cpu() - sum_w).abs().max()) <= 1e-5 * float(du_w[:E, :H].abs().sum(0).max())
    assert ops.segmax_bn_relu_backward(Mat.of(doutd, 0, H), argd, Mat.of(Zd, 0, H), dev(csr.rowptr), dev(csr.dst), dev(mean), dev(rstd),
                                       dev(gamma), dev(sdz), dev(sdzx), Mat.of(du, 0, H)) is None
    # (2b) the first edge layer with its statistics from the same pass == the layer, then col_stats over its live rows
    AB = torch.randn(n, 2 * ld, generator=g).to(DEV)
    z_a, z_b = torch.full((cap, ld), 9.0, device=DEV), torch.full((cap, ld), 9.0, device=DEV)
    assert ops.edge_gather_relu(Mat.of(AB, 0, H), Mat.of(AB, ld, H), csr_d, Mat.of(z_a, 0, H)) is None
    m_f, v_f, c_f = ops.edge_gather_relu(Mat.of(AB, 0, H), Mat.of(AB, ld, H), csr_d, Mat.of(z_b, 0, H), want_stats=True)
    assert torch.equal(z_a, z_b) and bool((z_a[:, H:] == 9.0).all()) and bool((z_a[E:] == 9.0).all())
    m_s, v_s, c_s = ops.col_stats(Mat.of(z_a, 0, H), rows_dev=csr_d.rowptr[n:n + 1])
    want_z = torch.relu(AB[:, :H].cpu()[csr.dst[:E].long()] + AB[:, ld:ld + H].cpu()[csr.src[:E].long()])
    assert torch.equal(z_a[:E, :H].cpu(), want_z) and float(c_f) == float(c_s) == float(E)
    assert torch.allclose(m_f, m_s, rtol=1e-6, atol=1e-7) and torch.allclose(v_f, v_s, rtol=1e-5, atol=1e-7)
    # (2c) the first layer's BatchNorm sums from M = du2^T Z1, db2 and W2 == the pass over dh = du2 W2 and Z1 (float64 reference)
    W2 = (torch.randn(H, H, generator=g) / H ** 0.5)
    du2 = du.clone(); du2[E:] = 0.0
    z1 = z_a.clone(); z1[E:] = 0.0
    rd = csr_d.rowptr[n:n + 1]
    m1, v1, _ = ops.col_stats(Mat.of(z1, 0, H), rows_dev=rd)
    r1 = torch.rsqrt(v1 + 1e-5)
    Mp = ops.gemm_tn(Mat.of(du2, 0, H), Mat.of(z1, 0, H), rows_dev=rd)
    got_a, got_b = ops.edge_bn_sums_from_products(Mp, sum_g, W2.to(DEV), m1, r1)
    dh64 = du2[:E, :H].double().cpu() @ W2.double()
    xh64 = (z1[:E, :H].double().cpu() - m1.double().cpu()) * r1.double().cpu()
    ref_a, ref_b = dh64.sum(0), (dh64 * xh64).sum(0)
    bound = dh64.norm(dim=0) * xh64.norm(dim=0) + 1e-30                      # Cauchy-Schwarz scale of each column's sum
    assert float(((got_b.cpu().double() - ref_b).abs() / bound).max()) <= 1e-4, float(((got_b.cpu().double() - ref_b).abs() / bound).max())
    assert float(((got_a.cpu().double() - ref_a).abs() / (dh64.norm(dim=0) * E ** 0.5 + 1e-30)).max()) <= 1e-5
    dhd = dh64.float().to(DEV)
    by_pass = ops.bn_backward_stats(Mat.of(dhd, 0, H), Mat.of(z1[:E].contiguous(), 0, H), m1, r1)
    assert float(((by_pass[1].cpu().double() - ref_b).abs() / bound).max()) <= 1e-5
    # (2d) the same sums with the contraction taken on rows CENTRED on the batch mean (morig_gemm_tn_shift; what edge_mlp's backward does):
    # Mc = du2^T (Z1 - 1 mean^T) needs no  - db2 (x) mean  afterwards, so nothing cancels -- also when mean >> std, the post-ReLU case
    # the uncentred product was weakest in: here every column of Z1 is lifted by 40 standard deviations
    for lift in (0.0, 40.0):
        z1l = z1.clone()
        z1l[:E, :H] += lift * z1[:E, :H].std(0)[None, :]
        ml, vl, _ = ops.col_stats(Mat.of(z1l, 0, H), rows_dev=rd)
        rl = torch.rsqrt(vl + 1e-5)
        xh = (z1l[:E, :H].double().cpu() - ml.double().cpu()) * rl.double().cpu()
        want_b = (dh64 * xh).sum(0)
        bnd = dh64.norm(dim=0) * xh.norm(dim=0) + 1e-30
        Mc = ops.gemm_tn(Mat.of(du2, 0, H), Mat.of(z1l, 0, H), rows_dev=rd, b_shift=ml.contiguous())
        _, cb = ops.edge_bn_sums_from_products(Mc, sum_g, W2.to(DEV), torch.zeros_like(ml), rl)
        err_c = float(((cb.cpu().double() - want_b).abs() / bnd).max())
        Mu = ops.gemm_tn(Mat.of(du2, 0, H), Mat.of(z1l, 0, H), rows_dev=rd)
        _, ub = ops.edge_bn_sums_from_products(Mu, sum_g, W2.to(DEV), ml, rl)
        err_u = float(((ub.cpu().double() - want_b).abs() / bnd).max())
        assert err_c <= 2e-5, (lift, err_c, err_u)                      # centred: float32-class whatever the mean
        if lift:
            assert err_u > 4 * err_c, (err_c, err_u)                    # (what the centring buys; the uncentred form degrades with the lift)
        back = (Mc + sum_g[:, None] * ml[None, :]).cpu().double()       # M = Mc + db2 (x) mean: the weight gradient's product
        m64 = du2[:E, :H].double().cpu().t() @ z1l[:E, :H].double().cpu()
        sc64 = float((du2[:E, :H].double().cpu().abs().t() @ z1l[:E, :H].double().cpu().abs()).max())
        assert float((back - m64).abs().max()) <= 1e-5 * sc64 and float((Mu.cpu().double() - m64).abs().max()) <= 1e-5 * sc64
    # (3) BatchNorm + ReLU inside the scatter sums; the plain scatter; both reproducible bit for bit
    Y = torch.relu(torch.randn(cap, ld, generator=g)); Y[E:] = float("nan")
    G = torch.randn(cap, ld, generator=g); G[E:] = float("nan")
    m1, r1 = Y[:E, :H].mean(0), 1.0 / (Y[:E, :H].std(0) + 0.1)
    k0, k1 = torch.randn(H, generator=g), torch.randn(H, generator=g)
    Yd, Gd = dev(Y), dev(G)
    for bn in (True, False):
        kw = dict(mean=m1, rstd=r1, gamma=gamma, sum_dz=k0, sum_dzx=k1) if bn else {}
        wa, wb = torch.zeros(n, H), torch.zeros(n, H)
        emu.edge_bn_scatter_backward(Mat.of(G.nan_to_num(0.0), 0, H), Mat.of(Y.nan_to_num(0.0), 0, H) if bn else None, csr, n, Mat.of(wa, 0, H),
                                     Mat.of(wb, 0, H), **kw)
        dab = torch.full((n, 2 * ld), 5.0, device=DEV)
        kwd = {k: dev(v) for k, v in kw.items()}
        ops.edge_bn_scatter_backward(Mat.of(Gd, 0, H), Mat.of(Yd, 0, H) if bn else None, csr_d, n, Mat.of(dab, 0, H), Mat.of(dab, ld, H), **kwd)
        sa = float(wa.abs().max())
        assert float((dab[:, :H].cpu() - wa).abs().max()) <= 2e-5 * sa and float((dab[:, ld:ld + H].cpu() - wb).abs().max()) <= 2e-5 * sa
        again = torch.full((n, 2 * ld), 5.0, device=DEV)
        ops.edge_bn_scatter_backward(Mat.of(Gd, 0, H), Mat.of(Yd, 0, H) if bn else None, csr_d, n, Mat.of(again, 0, H), Mat.of(again, ld, H), **kwd)
        assert torch.equal(dab, again)
        if bn:                                  # Y rebuilt from its operands (what edge_gather_relu stored: z_a above) == Y read back, bit for bit
            outs = []
            for rebuilt in (False, True):
                o = torch.full((n, 2 * ld), 5.0, device=DEV)
                ops.edge_bn_scatter_backward(Mat.of(Gd, 0, H), None if rebuilt else Mat.of(z_a, 0, H), csr_d, n, Mat.of(o, 0, H), Mat.of(o, ld, H),
                                             ZA=Mat.of(AB, 0, H) if rebuilt else None, ZB=Mat.of(AB, ld, H) if rebuilt else None, **kwd)
                outs.append(o)
            assert torch.equal(outs[0], outs[1])
        if not bn:                                                              # the atomic operator it replaces computes the same sums
            old = torch.full((n, 2 * ld), 5.0, device=DEV)
            ops.edge_scatter_backward(Mat.of(Gd, 0, H), csr_d, n, Mat.of(old, 0, H), Mat.of(old, ld, H))
            assert torch.allclose(old, dab, rtol=1e-4, atol=1e-5 * sa)


def _net_case(n_side=10, n_mesh=2, seed=5):
    batch = synth.make_batch(range(seed, seed + n_mesh), n_side=n_side, with_skin=False)
    return batch


def _grad_report(tag, mine, ref64, ref32, cos_floor, med_factor=5.0):
    """Whole-network gradients in fp32 are ill-conditioned at any size a test can afford: torch's own float32 autograd deviates
    from its float64 run by 1e-2 (median over the parameter tensors) to 1e-1 (worst tensor) of a tensor's scale -- batch statistics
    over a few hundred rows and arg-max near-ties that route a gradient to a different edge (measured: tools/debug_bw.py). So the
    criterion is statistical, with the float32 reference as the yardstick: (1) the median relative error over the parameter
    tensors is within 5x the float32 reference's, (2) every tensor points the same way as the float64 gradient (cosine), about as
    well as the float32 reference's does. The blocks themselves are held to 2e-4 above."""
    errs, errs32, worst_cos, worst_cos32 = [], [], 1.0, 1.0
    for (k, p), (_, q64), (_, q32) in zip(mine.named_parameters(), ref64.named_parameters(), ref32.named_parameters()):
        assert p.grad is not None, k
        a, r, r32 = p.grad.detach().cpu().double().flatten(), q64.grad.flatten(), q32.grad.double().flatten()
        scale = max(float(r.abs().max()), 1e-12)
        err = float((a - r).abs().max()) / scale
        cos = float(torch.dot(a, r) / (a.norm() * r.norm() + 1e-300))
        cos32 = float(torch.dot(r32, r) / (r32.norm() * r.norm() + 1e-300))
        errs.append(err); errs32.append(float((r32 - r).abs().max()) / scale)
        worst_cos, worst_cos32 = min(worst_cos, cos), min(worst_cos32, cos32)
        PARITY_LOG.append((f"backward:{tag}:{k}", err * scale, scale, err))
        assert cos >= min(cos_floor, cos32 - 0.02), (k, cos, cos32)
    med, med32 = sorted(errs)[len(errs) // 2], sorted(errs32)[len(errs32) // 2]
    assert med <= max(1e-2, med_factor * med32), (med, med32)
    return med, med32, worst_cos, worst_cos32


_ORACLE_RUNS = {}


def _oracle_once(key, run):
    """memo of a CPU-oracle run (networks with their gradients filled, outputs, initial state) shared by the parametrisations of a test"""
    if key not in _ORACLE_RUNS:
        _ORACLE_RUNS[key] = run()
    return _ORACLE_RUNS[key]


@pytest.fixture(params=["f32", "f16x3"])
def precision(request, monkeypatch):
    """network-level runs on both arithmetic paths of the train-mode forward: the default exact-fp32 MFMA contractions and the
    opt-in split-fp16 ones (MORIG_TRAIN_PRECISION, morig_amd/train_backward.py: ~2 bits less per product, which these
    ill-conditioned training forwards amplify -- hence the wider bands for it below)"""
    monkeypatch.setenv("MORIG_TRAIN_PRECISION", request.param)
    return request.param


def test_gcnrig_backward(precision):
    """one GCNRig (3 GCUMotion units = 12 edge MLPs, pooling, the transform MLP) end to end"""
    kw = dict(chn_feature=3, chn_output=32)
    b = _net_case()
    g = torch.Generator().manual_seed(1)
    feat = torch.randn(b.pos.shape[0], 3, generator=g) * 0.05
    w = torch.randn(b.pos.shape[0], 32, generator=g)

    def oracle_run():
        ref64 = _randomise(nets.RigGCN(**kw), 3).train().double()
        ref32 = copy.deepcopy(ref64).float()
        sd0 = copy.deepcopy(ref32.state_dict())                 # before the train-mode forwards move the BatchNorm buffers
        outs = {}
        for name, net, dt in (("r64", ref64, torch.float64), ("r32", ref32, torch.float32)):
            o = net(b.pos.to(dt), feat.to(dt), b.tpl_edge_index, b.geo_edge_index, b.batch)
            (o * w.to(dt)).sum().backward()
            outs[name] = o.detach()
        return ref64, ref32, sd0, outs
    ref64, ref32, sd0, outs = _oracle_once("gcnrig", oracle_run)
    mine = models.rignet.GCNRig(**kw).train()
    mine.load_state_dict(copy.deepcopy(sd0))
    mine.to(DEV)
    bd = b.to(DEV)
    st = TB.graph_state(bd)
    o = TB.gcnrig(mine, bd.pos.float(), feat.to(DEV), st["csr_tpl"], st["csr_geo"], st["batch"], st["mesh_ptr"], st["ng"])
    (o * w.to(DEV)).sum().backward()
    scale = float(outs["r64"].abs().max())
    err = float((o.detach().cpu().double() - outs["r64"]).abs().max()) / scale
    assert err <= max(1e-4, 4.0 * float((outs["r32"].double() - outs["r64"]).abs().max()) / scale), err
    _grad_report(f"gcnrig_{precision}", mine, ref64, ref32, 0.995 if precision == "f32" else 0.97, 5.0 if precision == "f32" else 15.0)


@pytest.mark.parametrize("aggr", ["attn", "mean"])
def test_jointnet_training_step_gradients(aggr, precision):
    """JointNetMotion in training mode: forward, a scalar loss over all three outputs, backward -- every parameter gradient
    against torch.autograd on the oracle (models/rignet.py:70-133, training/train_rig.py:136-195)."""
    kw = dict(num_keyframes=5, chn_output=3, aggr_method=aggr)
    b = _net_case(n_side=9, n_mesh=2, seed=11)
    g = torch.Generator().manual_seed(2)
    n = b.pos.shape[0]
    w_all, w_aggr, w_out = torch.randn(n, 5, 32, generator=g), torch.randn(n, 64 if aggr == "attn" else 32, generator=g), torch.randn(n, 3, generator=g)

    def loss(o, dt, dev="cpu"):
        return (o[0] * w_all.to(dev, dt)).sum() + (o[1] * w_aggr.to(dev, dt)).sum() + (o[2] * w_out.to(dev, dt)).sum()

    def oracle_run():
        ref64 = _randomise(nets.jointnet_motion(**kw), 7).train().double()
        ref32 = copy.deepcopy(ref64).float()
        sd0 = copy.deepcopy(ref32.state_dict())                 # before the train-mode forwards move the BatchNorm buffers
        outs = {}
        for name, net, dt in (("r64", ref64, torch.float64), ("r32", ref32, torch.float32)):
            bb = copy.copy(b)
            bb.pos = b.pos.to(dt)
            o = net(bb, b.pred_flow.to(dt))
            loss(o, dt).backward()
            outs[name] = [t.detach() for t in o]
        return ref64, ref32, sd0, outs
    # the float64 / float32 autograd runs of the CPU oracle do not depend on the HIP path's arithmetic: once per `aggr`, shared by
    # both `precision` cases (they were half of this test's host time)
    ref64, ref32, sd0, outs = _oracle_once(("jointnet", aggr), oracle_run)
    mine = models.jointnet_motion(**kw).train()
    mine.load_state_dict(copy.deepcopy(sd0))
    mine.to(DEV)
    bd = b.to(DEV)
    o = TB.motion_head_step(mine, bd, bd.pred_flow)
    loss(o, torch.float32, DEV).backward()
    for i, nm in enumerate(("motion_all", "motion_aggr", "out")):
        scale = max(float(outs["r64"][i].abs().max()), 1e-6)
        err = float((o[i].detach().cpu().double() - outs["r64"][i]).abs().max()) / scale
        slack = 4.0 * float((outs["r32"][i].double() - outs["r64"][i]).abs().max()) / scale
        PARITY_LOG.append((f"backward:jointnet_{aggr}:{nm}", err * scale, scale, err))
        assert err <= max(2e-4, slack * (1.0 if precision == "f32" else 4.0)), (nm, err, slack)
    _grad_report(f"jointnet_{aggr}_{precision}", mine, ref64, ref32, 0.99 if precision == "f32" else 0.95, 5.0 if precision == "f32" else 15.0)
    # BatchNorm running buffers moved exactly as the reference's did (5 motionNet passes + the head)
    for (k, v), (_, r) in zip(mine.state_dict().items(), ref32.state_dict().items()):
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(r), k


def test_module_api_trains_like_the_reference_loop(precision):
    """the reference's training loop shape (training/train_rig.py:136-195) through the module API, for the mask and skin heads:
    model.train(); out = model(data, flow); loss.backward(); optimizer.step(). Checked: the first loss equals the oracle's; ONE
    SGD step sized for a 0.5 % first-order decrease (lr = 0.005 loss / |g|^2) lowers the loss by that much to within a factor
    of three -- forward and backward agree with each other along the gradient; every parameter has a finite gradient and moves
    (unless |lr g| is below its last bit); BatchNorm buffers updated.
    (Multi-step trajectories are not compared: with these random BatchNorm gains the loss of the ORACLE moves by +-10 % per
    step at any usable learning rate, and two fp32 runs diverge after one step.)"""
    for arch, kw, width in (("masknet_motion", dict(num_keyframes=5, chn_output=1, aggr_method="attn"), 1),
                            ("skinnet_motion", dict(nearest_bone=5, use_Dg=False, use_Lf=False, num_keyframes=5, use_motion=True,
                                                    motion_dim=32), 5)):
        ref = _randomise(getattr(nets, arch)(**kw), 9).train()
        mine = getattr(models, arch)(**kw).train()
        mine.load_state_dict(copy.deepcopy(ref.state_dict()))
        mine.to(DEV)
        b = synth.make_batch(range(21, 23), n_side=9, with_skin=True)
        g = torch.Generator().manual_seed(4)
        target = torch.randn(b.pos.shape[0], width, generator=g)
        bd = b.to(DEV)
        want = float(((ref(b, b.pred_flow)[2] - target) ** 2).mean())
        before = [p.detach().clone() for p in mine.parameters()]

        def loss_of():
            return ((mine(bd, bd.pred_flow)[2] - target.to(DEV)) ** 2).mean()

        loss0 = loss_of()
        loss0.backward()
        assert abs(float(loss0) - want) <= 2e-3 * max(1.0, abs(want)), (arch, float(loss0), want)
        g2 = sum(float((p.grad.double() ** 2).sum()) for p in mine.parameters())
        lr = 0.005 * float(loss0) / g2
        opt = torch.optim.SGD(mine.parameters(), lr=lr)
        opt.step()
        with torch.no_grad():
            loss1 = float(loss_of())
        drop, expect = float(loss0) - loss1, lr * g2
        PARITY_LOG.append((f"backward:descent:{arch}_{precision}", abs(drop - expect), expect, abs(drop - expect) / max(expect, 1e-12)))
        assert 0.3 * expect <= drop <= 3.0 * expect, (arch, float(loss0), loss1, expect)
        assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in mine.parameters()), arch
        assert all(bool((p.grad != 0).any()) for p in mine.parameters()), arch      # every parameter takes part
        moved = [not torch.equal(p.detach(), q) for p, q in zip(mine.parameters(), before)]
        assert sum(moved) >= len(moved) // 3, (arch, sum(moved), len(moved))        # (|lr g| is below the last bit of many at this lr)
        nb = [v for k, v in mine.state_dict().items() if k.endswith("num_batches_tracked")]
        assert nb and all(int(v) > 0 for v in nb)


def test_skinnet_training_step_gradients(monkeypatch):
    """SkinMotion in training mode (models/rignet.py:136-205: three 256-wide GCUMotion units over [pos | skin samples] with 64-wide
    position branches, pooling, the classification head): every parameter gradient against torch.autograd on the oracle."""
    monkeypatch.setenv("MORIG_TRAIN_PRECISION", "f32")
    kw = dict(nearest_bone=5, use_Dg=False, use_Lf=False, num_keyframes=5, use_motion=True, motion_dim=32)
    ref64 = _randomise(nets.skinnet_motion(**kw), 13).train().double()
    ref32 = copy.deepcopy(ref64).float()
    mine = models.skinnet_motion(**kw).train()
    mine.load_state_dict(copy.deepcopy(ref32.state_dict()))
    mine.to(DEV)
    b = synth.make_batch(range(31, 33), n_side=9, with_skin=True)
    g = torch.Generator().manual_seed(6)
    n = b.pos.shape[0]
    w = [torch.randn(n, 5, 32, generator=g), torch.randn(n, 32, generator=g), torch.randn(n, 5, generator=g)]

    def loss(o, dt, dev="cpu"):
        return sum((o[i] * w[i].to(dev, dt)).sum() for i in range(3))

    outs = {}
    for name, net, dt in (("r64", ref64, torch.float64), ("r32", ref32, torch.float32)):
        bb = copy.copy(b)
        bb.pos, bb.skin_input = b.pos.to(dt), b.skin_input.to(dt)
        o = net(bb, b.pred_flow.to(dt))
        loss(o, dt).backward()
        outs[name] = [t.detach() for t in o]
    bd = b.to(DEV)
    o = mine(bd, bd.pred_flow)                       # module API: train mode + grad enabled -> the autograd blocks
    loss(o, torch.float32, DEV).backward()
    for i, nm in enumerate(("motion_all", "motion_aggr", "skin_cls_pred")):
        scale = max(float(outs["r64"][i].abs().max()), 1e-6)
        err = float((o[i].detach().cpu().double() - outs["r64"][i]).abs().max()) / scale
        slack = 4.0 * float((outs["r32"][i].double() - outs["r64"][i]).abs().max()) / scale
        PARITY_LOG.append((f"backward:skinnet:{nm}", err * scale, scale, err))
        assert err <= max(2e-4, slack), (nm, err, slack)
    _grad_report("skinnet_f32", mine, ref64, ref32, 0.99)


@pytest.mark.parametrize("vismask", [True, False])
def test_corrnet_training_step_gradients(vismask):
    """CorrNet in model.train() on the HIP operators (morig_amd/train_corr.py; training/train_corr_pose.py:61-70): forward against
    the float64 oracle with torch's own float32 run as the yardstick, every parameter's gradient against torch.autograd"""
    from helpers import check_corrnet_training
    check_corrnet_training(DEV, vismask)


def test_deformnet_training_step_gradients():
    """DeformNet in model.train() (training/train_deform_pose.py:29-40, 149-153): extractor frozen as the reference trains it,
    then nothing frozen"""
    from helpers import check_deformnet_training
    check_deformnet_training(DEV)


def test_training_step_is_bit_reproducible():
    """round 4: no atomics are left in the backward and the CSR segments are put into a canonical order (`canonical_csr`), so two
    training steps from the same state -- graphs rebuilt, every buffer reallocated -- give the same bits: outputs, every
    parameter's gradient, every running buffer"""
    from morig_amd.models import rignet as rn
    torch.manual_seed(5)
    net0 = _randomise(rn.GCNRig(chn_feature=3, chn_output=16), 9).train()
    b = synth.make_batch(range(3), n_side=12, with_skin=False).to(DEV)
    g = torch.Generator().manual_seed(1)
    feat = (torch.randn(b.pos.shape[0], 3, generator=g) * 0.05).to(DEV)
    w = torch.randn(b.pos.shape[0], 16, generator=g).to(DEV)

    def step(junk):
        net = copy.deepcopy(net0).to(DEV)
        scratch = [torch.empty(junk, device=DEV) for _ in range(3)]                # shifts the allocator's addresses between the runs
        st = TB.graph_state(b)
        out = TB.gcnrig(net, b.pos.float(), feat, st["csr_tpl"], st["csr_geo"], st["batch"], st["mesh_ptr"], st["ng"])
        (out * w).sum().backward()
        del scratch
        return out.detach().clone(), [p.grad.clone() for p in net.parameters()], [v.clone() for v in net.buffers()]

    o1, g1, b1 = step(1000)
    o2, g2, b2 = step(777777)
    assert torch.equal(o1, o2)
    for (k, _), a, c in zip(net0.named_parameters(), g1, g2):
        assert torch.equal(a, c), (k, float((a - c).abs().max()))
    for a, c in zip(b1, b2):
        assert torch.equal(a, c)


def test_jointnet_training_step_is_bit_reproducible():
    """the same through the module API: `model.train(); model(data, flow); loss.backward()` of jointnet_motion (keyframe loop,
    normalisation, CLS attention, head) twice from one state"""
    batch = synth.make_batch(range(2), n_side=10, with_skin=False).to(DEV)
    m0 = models.jointnet_motion(num_keyframes=5, chn_output=3, aggr_method="attn").train()
    synth.load_recipe(m0, 0, mild=True)

    def step():
        m = copy.deepcopy(m0).to(DEV)
        o = m(batch, batch.pred_flow)
        ((o[2] ** 2).mean() + (o[1] ** 2).mean()).backward()
        return [x.detach().clone() for x in o], [p.grad.clone() for p in m.parameters()]

    (o1, g1), (o2, g2) = step(), step()
    assert all(torch.equal(a, c) for a, c in zip(o1, o2))
    for (k, _), a, c in zip(m0.named_parameters(), g1, g2):
        assert torch.equal(a, c), (k, float((a - c).abs().max()))
