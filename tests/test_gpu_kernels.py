"""GPU: every C-ABI entry point of libmorig_hip.so against a torch restatement of what the kernel is
specified to compute (tests/emulate.py, run on CPU on the same seeded inputs). Integer/index results
must be bit-exact; fp32 contractions within 2e-5 * scale (different summation order only)."""
import pytest
import torch

from emulate import EmuOps
from helpers import maxdiff
from morig_amd import native, packing
from morig_amd.native import Mat

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module", params=["f16x3", "f32"])
def ops(request):
    """every kernel test runs on both arithmetic paths: split-fp16 MFMA (default) and fp32 MFMA."""
    from morig_amd import native
    o = native.get_ops()
    prev = o.precision
    o.precision = request.param
    yield o
    o.precision = prev


@pytest.fixture(autouse=True)
def _restore_precision():
    """several tests below pin the process-wide op layer to one arithmetic path: whatever a test sets is undone behind it, so the tests that
    follow run on the path their own fixture selected (order-dependent results otherwise)"""
    o = native.get_ops()
    prev = o.precision
    yield
    o.precision = prev


def _rand_graph(n, e, seed, hub=None):
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(0, n, (e,), generator=g)
    dst = torch.randint(0, n, (e,), generator=g)
    if hub is not None:                       # one target with a huge in-degree (segment spans several tiles)
        k = e // 3
        dst[:k] = hub
    src[::7] = dst[::7]                       # self loops to strip
    return torch.stack([src, dst])


def _segments(csr, E):
    rp = csr.rowptr.cpu().long()
    src = csr.src.cpu()[:E].long()
    return [sorted(src[rp[i]:rp[i + 1]].tolist()) for i in range(csr.n_nodes)]


@pytest.mark.parametrize("pad4", [False, True])
@pytest.mark.parametrize("n,e,hub", [(50, 400, None), (3000, 40000, 17), (5000, 0, None), (1, 5, None)])
def test_csr_build(ops, n, e, hub, pad4):
    ei = _rand_graph(n, e, 3, hub) if e else torch.zeros((2, 0), dtype=torch.long)
    ref = EmuOps().csr_build(ei, n, pad4=pad4) if e else None
    got = ops.csr_build(ei.to(DEV), n, pad4=pad4)
    torch.cuda.synchronize()
    assert int(got.status.item()) == 0
    if e == 0:
        k = 4 if pad4 else 1
        assert got.rowptr.cpu().tolist() == list(range(0, k * (n + 1), k))
        assert got.src.cpu()[:k * n].tolist() == [i // k for i in range(k * n)]
        return
    assert torch.equal(got.rowptr.cpu(), ref.rowptr)
    E = int(ref.rowptr[-1])
    assert torch.equal(got.dst.cpu()[:E], ref.dst[:E])
    assert _segments(got, E) == _segments(ref, E)
    if pad4:
        assert int((got.rowptr.cpu() % 4).abs().sum()) == 0


@pytest.mark.parametrize("n,e,hub", [(50, 300, None), (1000, 9000, 3), (4097, 30000, None), (7, 0, None), (2500, 40000, 11)])
def test_csr_build_dual(ops, n, e, hub):
    """morig_csr_build_dual: the plain and the 4-aligned CSR of one graph from one pass -- each equal to its own single build
    (same rowptr, same destinations, same multiset of sources per segment), out-of-range indices flagged once for both"""
    ei = _rand_graph(n, e, 5, hub) if e else torch.zeros((2, 0), dtype=torch.long)
    a, b = ops.csr_build_dual(ei.to(DEV), n, min4=False)
    torch.cuda.synchronize()
    assert int(a.status.item()) == 0 and b.quad and not a.quad
    for got, pad4 in ((a, False), (b, True)):
        want = ops.csr_build(ei.to(DEV), n, pad4=pad4)
        assert torch.equal(got.rowptr, want.rowptr)
        E = int(want.rowptr[-1])
        assert torch.equal(got.dst[:E], want.dst[:E])
        assert _segments(got, E) == _segments(want, E)
    bad = ei.clone() if e else torch.tensor([[0], [0]])
    bad[0, 0] = n + 5
    a2, _ = ops.csr_build_dual(bad.to(DEV), n)
    torch.cuda.synchronize()
    assert int(a2.status.item()) != 0


def test_csr_build_flags_bad_index(ops):
    ei = torch.tensor([[0, 1, 9], [1, 2, 0]])
    got = ops.csr_build(ei.to(DEV), 3)
    torch.cuda.synchronize()
    assert int(got.status.item()) != 0


def _lin(N, K, seed, bn=True):
    g = torch.Generator().manual_seed(seed)
    W = torch.randn(N, K, generator=g) / (K ** 0.5)
    b = torch.randn(N, generator=g) * 0.1
    p = packing.pack_linear(W, b)
    if bn:
        Np = p.W.shape[0]
        p.scale = torch.randn(Np, generator=g)
        p.shift = torch.randn(Np, generator=g) * 0.2
    return p


GEMM_CASES = [
    # M, N, K, relu, bn
    (1, 3, 3, False, False), (130, 32, 64, True, True), (257, 64, 33, True, True), (1000, 100, 835, True, True),
    (513, 128, 832, False, False), (300, 1024, 1859, True, True), (128, 256, 1024, True, True), (77, 5, 512, False, False),
    (4096, 512, 544, True, True),
]


@pytest.mark.parametrize("M,N,K,relu,bn", GEMM_CASES)
def test_gemm_store(ops, M, N, K, relu, bn):
    g = torch.Generator().manual_seed(M + N + K)
    ld = (K + 3) // 4 * 4 + 8
    xb = torch.randn(M, ld, generator=g)
    xb[:, 4 + K:] = float("nan")                          # anything beyond K must never be read into the sum
    lin = _lin(N, K, 11, bn)
    yb_ref = torch.zeros(M, N + 5)
    EmuOps().gemm(Mat.of(xb, 4, K), lin, relu, Y=Mat.of(yb_ref, 2, N))
    xg, ling = xb.to(DEV), packing.to_device(lin, DEV)
    yb = torch.zeros(M, N + 5, device=DEV)
    ops.gemm(Mat.of(xg, 4, K), ling, relu, Y=Mat.of(yb, 2, N))
    torch.cuda.synchronize()
    scale = max(1.0, yb_ref.abs().max().item())
    assert maxdiff(yb, yb_ref) <= 2e-5 * scale
    assert float(yb[:, :2].abs().sum()) == 0 and float(yb[:, 2 + N:].abs().sum()) == 0     # window respected


@pytest.mark.parametrize("n,T,C,heads", [(1003, 5, 32, 2), (7, 1, 32, 1), (4099, 8, 32, 4), (70, 5, 32, 3), (130, 5, 16, 2), (65, 2, 48, 2)])
def test_cls_attention_register_and_lds_forms(ops, n, T, C, heads):
    """morig_cls_attention (models/rignet.py:28-43, CLS row only): the register-resident form for C = 32 [r06] (8 lanes per vertex, 16-byte
    loads / stores, vertex counts that are not multiples of 8, 1 ... 4 heads, 1 ... 8 frames) and the LDS-staged form for other widths,
    both against the float32 emulation of the reference's softmax(QK^T)V"""
    g = torch.Generator().manual_seed(n + T + heads)
    xa = torch.nn.functional.normalize(torch.randn(n, T, C, generator=g), dim=2) * 3.0
    gq, cls = torch.randn(heads, C, generator=g), torch.randn(C, generator=g)
    a_ref, a = torch.zeros(n, heads * C + 4), torch.full((n, heads * C + 4), 7.0, device=DEV)
    EmuOps().cls_attention(xa, gq, cls, Mat.of(a_ref, 0, heads * C))
    ops.cls_attention(xa.to(DEV), gq.to(DEV), cls.to(DEV), Mat.of(a, 0, heads * C))
    torch.cuda.synchronize()
    assert maxdiff(a[:, :heads * C], a_ref[:, :heads * C]) <= 3e-6
    assert bool((a[:, heads * C:] == 7.0).all())                       # the window is respected


@pytest.mark.parametrize("M,N,K,relu,bn", [(1, 1024, 1024, False, False), (2, 1024, 1024, True, True), (10, 1024, 1024, False, True),
                                          (64, 1024, 1024, False, False), (128, 1024, 1024, True, True), (129, 1024, 1024, True, True), (100, 300, 260, True, True),
                                          (33, 130, 128, False, False), (7, 1024, 512, True, False), (5, 131, 1024, False, True)])
def test_gemm_on_a_few_rows(ops, M, N, K, relu, bn):
    """[r06] morig_gemm with at most 128 rows of fp32 X (the per-mesh vectors: the Linear behind the pooled global feature, models/rignet.py:
    60-63) runs on vertex_ops.hip's few-rows kernel -- weights spread over N / 4 workgroups, fp32 FMAs, a fixed reduction order: float32-class
    results (against float64), the column window respected, nothing beyond K read, bit-identical from run to run, the same in both modes"""
    g = torch.Generator().manual_seed(M + N + K)
    ld = K + 8
    xb = torch.randn(M, ld, generator=g)
    xb[:, 4 + K:] = float("nan")
    xb[:, :4] = float("nan")
    lin = _lin(N, K, 13, bn)
    ref = xb[:, 4:4 + K].double() @ lin.W[:N, :K].double().t() + lin.bias[:N].double()
    if relu:
        ref = ref.clamp_min(0.0)
    if lin.scale is not None:
        ref = ref * lin.scale[:N].double() + lin.shift[:N].double()
    xg, ling = xb.to(DEV), packing.to_device(lin, DEV)
    outs = []
    for _ in range(2):
        yb = torch.zeros(M, N + 5, device=DEV)
        ops.gemm(Mat.of(xg, 4, K), ling, relu, Y=Mat.of(yb, 2, N))
        torch.cuda.synchronize()
        outs.append(yb)
    assert torch.equal(outs[0], outs[1])
    yb = outs[0].cpu()
    scale = max(1.0, ref.abs().max().item())
    assert float((yb[:, 2:2 + N].double() - ref).abs().max()) <= 3e-6 * scale
    assert float(yb[:, :2].abs().sum()) == 0 and float(yb[:, 2 + N:].abs().sum()) == 0


@pytest.mark.parametrize("M,N,K,scale", [(1000, 256, 864, 1.0), (513, 64, 300, 1.0e3), (700, 1024, 256, 1.0e-3), (300, 32, 64, 3.0e4)])
def test_exact_path_on_bf16x6_is_float32_class(M, N, K, scale):
    """The exact path's default arithmetic [r06] (MORIG_SPLIT_BF16X6: both fp32 operands split into three bf16 limbs in the kernel, six
    MFMAs per product) against float64, beside v_mfma_f32_32x32x2_f32 on the same inputs: its error is of the fp32-MFMA kernel's size (the
    dropped limb products are <= 2^-24 of a product), at every magnitude -- bf16 keeps float32's exponent range, so inputs of 3e4 x 1e3 need no
    range guard and inputs of 1e-3 lose nothing to subnormals."""
    o = native.get_ops()
    o.precision = "f32"
    prev = o.exact_arith
    g = torch.Generator().manual_seed(M + N)
    x = (torch.randn(M, (K + 3) // 4 * 4, generator=g) * scale)
    lin = _lin(N, K, 4, bn=False)
    lin.W = lin.W * (1.0e3 if scale > 1.0e4 else 1.0)
    ref = (x[:, :K].double() @ lin.W[:N, :K].double().t() + lin.bias[:N].double())
    if lin.scale is not None:                      # (row-normalised pack: the power-of-two factors are undone through scale)
        ref = ref * lin.scale[:N].double() + lin.shift[:N].double()
    errs = {}
    try:
        for mode in ("f32", "bf16x6"):
            o.exact_arith = mode
            y = torch.zeros(M, N, device=DEV)
            o.gemm(Mat.of(x.to(DEV), 0, K), packing.to_device(lin, DEV), False, Y=Mat.of(y))
            torch.cuda.synchronize()
            errs[mode] = (y.cpu().double() - ref).abs().max().item()
    finally:
        o.exact_arith = prev
    mag = ref.abs().max().item()
    assert errs["bf16x6"] <= 4.0 * errs["f32"] + 1e-7 * mag, (errs, mag)
    assert errs["bf16x6"] <= 3e-6 * mag, (errs, mag)              # float32-class: a K-term sum of float32 products


def test_gemm_refuses_a_split_image_without_its_range_guard():
    """the weight format of morig_gemm is explicit (w_split_format): an fp16-split image handed over WITHOUT the overflow word is
    MORIG_E_INVALID, not silently taken for the bf16 split; an unknown format value is refused as well"""
    import ctypes as C
    o = native.get_ops()
    lin = packing.to_device(_lin(64, 64, 5, False), DEV)
    assert lin.Wsplit is not None
    x = torch.randn(128, 64, device=DEV)
    y = torch.zeros(128, 64, device=DEV)

    def call(overflow, fmt):
        a = native._args(native.GemmArgs)
        a.M, a.N, a.K = 128, 64, 64
        a.X, a.ldx = x.data_ptr(), 64
        a.W, a.ldw = lin.W.data_ptr(), lin.W.stride(0)
        a.Y, a.ldy = y.data_ptr(), 64
        a.W_split, a.overflow, a.w_split_format = lin.Wsplit.data_ptr(), overflow, fmt
        return o.lib.morig_gemm(C.byref(a), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    assert call(flag.data_ptr(), 0) == 0
    assert call(0, 0) == -1 and call(flag.data_ptr(), 7) == -1          # MORIG_E_INVALID
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,N,K,nseg", [(1000, 1024, 832, 7), (130, 512, 256, 130), (4096, 1024, 64, 1), (300, 200, 100, 3)])
def test_gemm_pool_and_rowbias(ops, M, N, K, nseg):
    g = torch.Generator().manual_seed(M + nseg)
    x = torch.randn(M, (K + 3) // 4 * 4, generator=g)
    seg = torch.sort(torch.randint(0, nseg, (M,), generator=g))[0].int()
    seg[0], seg[-1] = 0, nseg - 1
    if nseg == 130:
        seg = torch.arange(M).int()
    present = torch.unique(seg.long())
    lin = _lin(N, K, 5)
    pool_ref = torch.zeros(nseg, N)
    EmuOps().gemm(Mat.of(x, 0, K), lin, True, seg=seg, pool=pool_ref)
    pool = torch.zeros(nseg, N, device=DEV)
    ops.gemm(Mat.of(x.to(DEV), 0, K), packing.to_device(lin, DEV), True, seg=seg.to(DEV), pool=pool)
    torch.cuda.synchronize()
    assert maxdiff(pool[present.to(DEV)], pool_ref[present]) <= 2e-5 * max(1.0, pool_ref[present].abs().max().item())
    # row bias indexed by segment
    rb = torch.randn(nseg, N, generator=g)
    y_ref = torch.zeros(M, N)
    EmuOps().gemm(Mat.of(x, 0, K), lin, True, Y=Mat.of(y_ref), rowbias=Mat.of(rb), seg=seg)
    y = torch.zeros(M, N, device=DEV)
    ops.gemm(Mat.of(x.to(DEV), 0, K), packing.to_device(lin, DEV), True, Y=Mat.of(y), rowbias=Mat.of(rb.to(DEV)), seg=seg.to(DEV))
    torch.cuda.synchronize()
    assert maxdiff(y, y_ref) <= 2e-5 * max(1.0, y_ref.abs().max().item())


def _edge_pack(H, seed, folded=False):
    """folded=True: hidden affine absent (s1 = t1 = None), the form morig_amd.packing produces and the only
    one the wave-specialised kernel (edge_pc.hip) takes."""
    g = torch.Generator().manual_seed(seed)
    Hp, Kp = max(H, 32), (H + 31) // 32 * 32
    W2 = torch.zeros(Hp, Kp)
    W2[:H, :H] = torch.randn(H, H, generator=g) / (H ** 0.5)

    def vec(n, fill, f):
        v = torch.full((n,), fill)
        v[:H] = f(H)
        return v
    rn = lambda k: torch.randn(k, generator=g)
    s1, t1 = vec(Kp, 1.0, rn), vec(Kp, 0.0, lambda k: rn(k) * 0.2)
    pe = packing.PackedEdge(H, s1, t1, W2, vec(Hp, 0.0, lambda k: rn(k) * 0.1), vec(Hp, 1.0, rn), vec(Hp, 0.0, lambda k: rn(k) * 0.2))
    if folded:
        pe.s1 = pe.t1 = None
    pe.W2split = packing.split_f16(W2) if H >= 32 else None
    return pe


@pytest.mark.parametrize("n,e,hub,reps,pad4", [(300, 2500, 5, 1, False), (1500, 9000, None, 3, False), (700, 5000, 3, 2, True), (5000, 60000, None, 1, True)])
def test_edgeconv_x3(ops, n, e, hub, reps, pad4):
    """morig_edgeconv_x3: a 32-wide EdgeConv on a 3-channel vertex input with the first Linear evaluated in the loader from the
    gathered endpoints -- against the emulation's [A | B] form (A = W1a x + b1, B = W1b x per vertex, then the usual layer)"""
    g = torch.Generator().manual_seed(n + reps)
    ei = _rand_graph(n, e, 7, hub)
    x = torch.zeros(reps * n, 4)
    x[:, :3] = torch.randn(reps * n, 3, generator=g)
    x[:, 3] = float("nan")                                      # column 3 must never enter the sum
    W1a, W1b, b1 = torch.zeros(32, 4), torch.zeros(32, 4), torch.randn(32, generator=g) * 0.2
    W1a[:, :3], W1b[:, :3] = torch.randn(32, 3, generator=g), torch.randn(32, 3, generator=g)
    pe = _edge_pack(32, 21, folded=True)
    emu = EmuOps()
    want = torch.zeros(reps * n, 40)
    emu.edgeconv_x3(Mat.of(x), (W1a, W1b, b1), emu.csr_build(ei, n, pad4=pad4), pe, Mat.of(want, 4, 32, 0, n), replicas=reps,
                    in_rep_stride=n, out_rep_stride=n)
    got = torch.zeros(reps * n, 40, device=DEV)
    csr = ops.csr_build(ei.to(DEV), n, pad4=pad4)
    ops.edgeconv_x3(Mat.of(x.to(DEV)), (W1a.to(DEV), W1b.to(DEV), b1.to(DEV)), csr, packing.to_device(pe, DEV),
                    Mat.of(got, 4, 32, 0, n), replicas=reps, in_rep_stride=n, out_rep_stride=n)
    torch.cuda.synchronize()
    assert maxdiff(got, want) <= 2e-5 * max(1.0, want.abs().max().item())
    again = torch.zeros(reps * n, 40, device=DEV)
    ops.edgeconv_x3(Mat.of(x.to(DEV)), (W1a.to(DEV), W1b.to(DEV), b1.to(DEV)), csr, packing.to_device(pe, DEV),
                    Mat.of(again, 4, 32, 0, n), replicas=reps, in_rep_stride=n, out_rep_stride=n)
    assert torch.equal(got, again)


@pytest.mark.parametrize("folded,pad4", [(False, False), (True, False), (True, True)])
@pytest.mark.parametrize("H,ld_extra", [(16, 3), (32, 3), (64, 3), (128, 3), (256, 3), (128, 4), (256, 4)])
@pytest.mark.parametrize("n,e,hub,reps,shared", [(300, 2500, 5, 1, False), (1500, 9000, None, 3, False), (700, 5000, 3, 2, True)])
def test_edgeconv(ops, H, ld_extra, n, e, hub, reps, shared, folded, pad4):
    """ld_extra = 4: 16-byte aligned output rows, the layout the persistent kernel (edge_pp.hip) needs; 3: misaligned
    rows, which H = 128 / 256 serve with the one-shot kernel (edge_pc.hip)."""
    g = torch.Generator().manual_seed(H + n)
    ei = _rand_graph(n, e, 9, hub)
    rows_in = n if shared else n * reps
    ab = torch.randn(rows_in, 2 * H + 4, generator=g)
    ec = _edge_pack(H, 21, folded)
    emu = EmuOps()
    csr_ref = emu.csr_build(ei, n)
    out_ref = torch.zeros(n * reps, H + ld_extra)
    emu.edgeconv(Mat.of(ab, 0, H), Mat.of(ab, H, H), csr_ref, ec, Mat.of(out_ref, 0, H), replicas=reps,
                 in_rep_stride=0 if shared else n, out_rep_stride=n)
    csr = ops.csr_build(ei.to(DEV), n, pad4=pad4)
    abg = ab.to(DEV)
    out = torch.zeros(n * reps, H + ld_extra, device=DEV)
    ops.edgeconv(Mat.of(abg, 0, H), Mat.of(abg, H, H), csr, packing.to_device(ec, DEV), Mat.of(out, 0, H), replicas=reps,
                 in_rep_stride=0 if shared else n, out_rep_stride=n)
    torch.cuda.synchronize()
    assert not torch.isnan(out).any()
    assert maxdiff(out, out_ref) <= 2e-5 * max(1.0, out_ref.abs().max().item())
    assert float(out[:, H:].abs().sum()) == 0


@pytest.mark.parametrize("pad4", [False, True])
@pytest.mark.parametrize("H", [128, 256])
def test_edgeconv_persistent_many_tiles(ops, H, pad4):
    """~20 tiles per workgroup of the persistent kernel (cross-tile prefetch, double-buffered segment ids, tile-straddling
    segments incl. one hub longer than several tiles), two replicas reading the same input."""
    n, e, reps = 20000, 300000, 2
    g = torch.Generator().manual_seed(H)
    ei = _rand_graph(n, e, 31, hub=777)
    extra = torch.stack([torch.randint(0, n, (700,), generator=g), torch.full((700,), 1234)])      # a 700-edge destination
    ei = torch.cat([ei, extra], dim=1)
    ab = torch.randn(n, 2 * H, generator=g)
    ec = _edge_pack(H, 5, folded=True)
    emu = EmuOps()
    out_ref = torch.zeros(n * reps, H)
    emu.edgeconv(Mat.of(ab, 0, H), Mat.of(ab, H, H), emu.csr_build(ei, n), ec, Mat.of(out_ref), replicas=reps,
                 in_rep_stride=0, out_rep_stride=n)
    csr = ops.csr_build(ei.to(DEV), n, pad4=pad4)
    abg = ab.to(DEV)
    out = torch.full((n * reps, H), float("nan"), device=DEV)
    ops.edgeconv(Mat.of(abg, 0, H), Mat.of(abg, H, H), csr, packing.to_device(ec, DEV), Mat.of(out), replicas=reps,
                 in_rep_stride=0, out_rep_stride=n)
    torch.cuda.synchronize()
    assert not torch.isnan(out).any()
    assert maxdiff(out, out_ref) <= 2e-5 * max(1.0, out_ref.abs().max().item())
    assert torch.equal(out[:n], out[n:])                      # replicas of one input: identical bits


@pytest.mark.parametrize("n,e,hub", [(50, 300, None), (3000, 2500, None), (4097, 30000, 7)])
def test_csr_build_min4_segments(ops, n, e, hub):
    """MORIG_CSR_MIN4 (single and dual build): the plain CSR with every segment filled up to 4 rows by copies of its self loop -- the
    same edges otherwise (the form the mixed-quad H = 256 EdgeConv kernel takes instead of 4-aligned segments)"""
    ei = _rand_graph(n, e, 9, hub)
    want = ops.csr_build(ei.to(DEV), n)
    single = ops.csr_build(ei.to(DEV), n, min4=True)
    dual, quad = ops.csr_build_dual(ei.to(DEV), n, min4=True)
    plain_dual, _ = ops.csr_build_dual(ei.to(DEV), n, min4=False)
    torch.cuda.synchronize()
    assert torch.equal(plain_dual.rowptr, want.rowptr) and not plain_dual.min4 and dual.min4 and single.min4 and quad.quad
    deg = (want.rowptr[1:] - want.rowptr[:-1]).cpu()
    for got in (single, dual):
        assert int(got.status.item()) == 0
        glen = (got.rowptr[1:] - got.rowptr[:-1]).cpu()
        assert torch.equal(glen, deg.clamp(min=4))
        E = int(got.rowptr[-1])
        segs, ref = _segments(got, E), _segments(want, int(want.rowptr[-1]))
        for v in range(0, n, max(1, n // 300)):
            extra = [v] * int(glen[v] - deg[v])
            assert sorted(segs[v]) == sorted(list(ref[v]) + extra), v


@pytest.mark.parametrize("H,split", [(256, True), (128, True), (256, False), (32, False), (16, False)])
@pytest.mark.parametrize("n,e1,e2,hub,reps", [(700, 5000, 9000, 3, 2), (20000, 120000, 300000, 777, 1), (64, 200, 90, None, 3)])
def test_edgeconv_pair_shares_the_boundary_passes(ops, H, split, n, e1, e2, hub, reps):
    """[r06] morig_edgeconv_args.init_with / split_with (native.edgeconv_pair): the two EdgeConvs of a unit -- two graphs, two column
    blocks of the same rows -- with ONE identity pass in front of both kernels and, for split rows, ONE conversion pass behind both:
    bit-identical to two plain calls, on the wide split-rows kernels, their fp32-rows form and the narrow tile engine, with hubs that
    straddle many tiles and graphs of different sizes (the shared launch covers the larger candidate count); a pair whose flags do not
    fit together is refused. The 3-channel form (morig_edgeconv_x3_args.init_with) likewise."""
    from morig_amd import native
    o = ops
    if split and o.precision != "f16x3":
        pytest.skip("split rows exist on the split-fp16 path only")
    g = torch.Generator().manual_seed(H + n)
    ei1, ei2 = _rand_graph(n, e1, 13, hub), _rand_graph(n, e2, 29, None if hub is None else hub + 5)
    Hc = max(H, 32)
    ab = torch.randn(n * reps, 4 * Hc, generator=g).to(DEV)
    ec1, ec2 = (packing.to_device(_edge_pack(H, sd, folded=True), DEV) for sd in (5, 6))
    wide = H >= 128
    csr1, csr2 = (o.csr_build(ei.to(DEV), n, pad4=wide) for ei in (ei1, ei2))
    ld = 2 * Hc + 64
    kw = dict(replicas=reps, in_rep_stride=n, out_rep_stride=n)
    c1 = dict(A=Mat.of(ab, 0, H), B=Mat.of(ab, Hc, H), csr=csr1, ec=ec1, **kw)
    c2 = dict(A=Mat.of(ab, 2 * Hc, H), B=Mat.of(ab, 3 * Hc, H), csr=csr2, ec=ec2, **kw)
    if split and not (o.edgeconv_can_split_out(out=Mat.of(torch.zeros(n * reps, ld, device=DEV), 0, H), **c1)):
        pytest.skip("this launch runs on a kernel without split rows (environment switch)")
    sep = torch.full((n * reps, ld), 7.0, device=DEV)
    o.edgeconv(out=Mat.of(sep, 0, H), out_split=split, **c1)
    o.edgeconv(out=Mat.of(sep, Hc, H), out_split=split, **c2)
    par = torch.full((n * reps, ld), 7.0, device=DEV)
    o.edgeconv_pair(dict(out=Mat.of(par, 0, H), out_split=split, **c1), dict(out=Mat.of(par, Hc, H), out_split=split, **c2))
    torch.cuda.synchronize()
    assert torch.equal(sep.view(torch.int32), par.view(torch.int32))
    assert not torch.isnan(par[:, :H]).any() or split          # (split rows are half pairs: NaN patterns are possible bit-wise, not as values)
    if split:
        assert not torch.isnan(packing.unsplit_f16(par[:, :2 * Hc].contiguous().cpu(), 2 * Hc)).any()
    # flags that do not fit together: a skipped pass without a partner that does it is the caller's business, but a partner pointer on
    # a launch that skips its own pass, or conversion flags on fp32 rows, are refused
    a = o._edge_args(c1["A"], c1["B"], csr1, ec1, Mat.of(par, 0, H), reps, n, n)
    b = o._edge_args(c2["A"], c2["B"], csr2, ec2, Mat.of(par, Hc, H), reps, n, n)
    import ctypes as C
    a.skip_init, a.init_with = 1, C.addressof(b)
    assert o.lib.morig_edgeconv(C.byref(a), native._stream()) == -1
    a.skip_init, a.init_with, a.skip_split = 0, None, 1          # fp32 rows have no conversion pass
    assert o.lib.morig_edgeconv(C.byref(a), native._stream()) == -1
    torch.cuda.synchronize()
    if H == 32 and o.precision == "f16x3":
        x = torch.zeros(n * reps, 4)
        x[:, :3] = torch.randn(n * reps, 3, generator=g)
        x = x.to(DEV)
        firsts = [tuple(t.to(DEV) for t in packing.pack_first_x3(torch.randn(32, 3, generator=g), torch.randn(32, 3, generator=g),
                                                                 torch.randn(32, generator=g))) for _ in range(2)]
        sep3 = torch.full((n * reps, ld), 7.0, device=DEV)
        o.edgeconv_x3(Mat.of(x), firsts[0], csr1, ec1, Mat.of(sep3, 0, 32), **kw)
        o.edgeconv_x3(Mat.of(x), firsts[1], csr2, ec2, Mat.of(sep3, 32, 32), **kw)
        par3 = torch.full((n * reps, ld), 7.0, device=DEV)
        o.edgeconv_x3_pair(dict(X=Mat.of(x), first=firsts[0], csr=csr1, ec=ec1, out=Mat.of(par3, 0, 32), **kw),
                           dict(X=Mat.of(x), first=firsts[1], csr=csr2, ec=ec2, out=Mat.of(par3, 32, 32), **kw))
        torch.cuda.synchronize()
        assert torch.equal(sep3, par3) and not torch.isnan(par3).any()


@pytest.mark.parametrize("H,kind", [(128, "pad4"), (256, "pad4"), (256, "min4")])
@pytest.mark.parametrize("n,e,hub,reps", [(700, 5000, 3, 2), (20000, 300000, 777, 2), (64, 200, None, 1), (3000, 2500, None, 3)])
def test_edgeconv_split_fp16_rows(H, kind, n, e, hub, reps):
    """morig_edgeconv out_split: the 4-aligned-CSR kernels store split-fp16 rows (chunk = [32 hi | 32 lo]) into a chunk-aligned column
    window -- whole segments from the kernel, tile-straddling ones (atomics in fp32) through the boundary pass, a hub longer than several
    tiles converted exactly once. Decoded, every element is the fp32 launch's result to the split's resolution (lo is fp16(v - hi):
    2^-22 relative, 2^-25 absolute where lo is an fp16 subnormal); the columns beside the window stay untouched; a result beyond the
    fp16 range raises the guard."""
    from morig_amd import native
    o = native.get_ops()
    o.precision = "f16x3"
    g = torch.Generator().manual_seed(H + n)
    ei = _rand_graph(n, e, 13, hub)
    if hub is not None and n >= 20000:
        ei = torch.cat([ei, torch.stack([torch.randint(0, n, (700,), generator=g), torch.full((700,), 1234)])], dim=1)
    ab = torch.randn(n * reps, 2 * H, generator=g).to(DEV)
    ec = packing.to_device(_edge_pack(H, 5, folded=True), DEV)
    ec.s2 = ec.s2 * torch.where(torch.arange(ec.s2.numel(), device=DEV) % 3 == 0, -1.0, 1.0)       # both signs of the BN scale
    # kind "min4" [r06]: segments of >= 4 rows, NOT 4-aligned -- the mixed-quad form of the H = 256 kernel (quads that straddle two
    # segments, mid-quad starts, a mixed first quad of a tile); its fp32 reference launch runs on the generic kernel (same CSR)
    csr = o.csr_build(ei.to(DEV), n, pad4=True) if kind == "pad4" else o.csr_build(ei.to(DEV), n, min4=True)
    ld = 2 * H + 64
    kw = dict(replicas=reps, in_rep_stride=n, out_rep_stride=n)
    plain = torch.zeros(n * reps, ld, device=DEV)
    o.edgeconv(Mat.of(ab, 0, H), Mat.of(ab, H, H), csr, ec, Mat.of(plain, H, H), **kw)
    rows = torch.full((n * reps, ld), 7.0, device=DEV)
    if not o.edgeconv_can_split_out(Mat.of(ab, 0, H), Mat.of(ab, H, H), csr, ec, Mat.of(rows, H, H), **kw):
        import os
        assert any(os.environ.get(k) for k in ("MORIG_RL128", "MORIG_EDGE_KERNEL", "MORIG_NO_EDGE_PC", "MORIG_EDGE_SPLIT_OUT", "MORIG_EDGE_MIX")), \
            "split rows refused without a kernel-selection switch in the environment"
        with pytest.raises(native.MorigNativeError):             # ... and the launch refuses them too
            o.edgeconv(Mat.of(ab, 0, H), Mat.of(ab, H, H), csr, ec, Mat.of(rows, H, H), out_split=True, **kw)
        pytest.skip("this launch runs on a kernel without split rows (environment switch)")
    o._flag(ab.device).zero_()
    o.edgeconv(Mat.of(ab, 0, H), Mat.of(ab, H, H), csr, ec, Mat.of(rows, H, H), out_split=True, **kw)
    torch.cuda.synchronize()
    assert int(o._flag(ab.device).item()) == 0
    got = packing.unsplit_f16(rows[:, H:2 * H].contiguous().cpu(), H)
    want = plain[:, H:2 * H].cpu()
    assert not torch.isnan(got).any()
    if kind == "pad4":      # same kernel, same accumulation order: only the split's resolution separates the two
        assert ((got - want).abs() <= 2.0 ** -21 * want.abs() + 2.0 ** -24).all(), maxdiff(got, want)
    else:                   # the fp32 reference launch of a MIN4 CSR runs on the generic kernel: another summation order
        assert maxdiff(got, want) <= 2e-5 * max(1.0, want.abs().max().item())
    assert float((rows[:, :H] - 7.0).abs().sum()) == 0 and float((rows[:, 2 * H:] - 7.0).abs().sum()) == 0
    # twice the same bits (no order dependence through the atomics + the boundary pass)
    again = torch.full((n * reps, ld), 7.0, device=DEV)
    o.edgeconv(Mat.of(ab, 0, H), Mat.of(ab, H, H), csr, ec, Mat.of(again, H, H), out_split=True, **kw)
    assert torch.equal(rows, again)
    # refused where the launch would not take one of the two kernels (plain CSR), and the guard: a result of 1e5
    csr1 = o.csr_build(ei.to(DEV), n)
    assert not o.edgeconv_can_split_out(Mat.of(ab, 0, H), Mat.of(ab, H, H), csr1, ec, Mat.of(rows, H, H), **kw)
    with pytest.raises(native.MorigNativeError):
        o.edgeconv(Mat.of(ab, 0, H), Mat.of(ab, H, H), csr1, ec, Mat.of(rows, H, H), out_split=True, **kw)
    ec.t2 = ec.t2 + 1.0e5
    o.edgeconv(Mat.of(ab, 0, H), Mat.of(ab, H, H), csr, ec, Mat.of(rows, H, H), out_split=True, **kw)
    torch.cuda.synchronize()
    assert int(o._flag(ab.device).item()) == 1
    o._flag(ab.device).zero_()


def test_edgeconv_sign_cases(ops):
    """negative BN scales (max of a decreasing function) and all-negative outputs (the integer-atomic
    max identity must not leak)."""
    H, n = 32, 400
    ei = _rand_graph(n, 6000, 4, hub=11)
    ec = _edge_pack(H, 2)
    ec.s2 = -ec.s2.abs()
    ec.t2 = ec.t2 - 5.0
    ab = torch.randn(n, 2 * H)
    emu = EmuOps()
    out_ref = torch.zeros(n, H)
    emu.edgeconv(Mat.of(ab, 0, H), Mat.of(ab, H, H), emu.csr_build(ei, n), ec, Mat.of(out_ref))
    out = torch.zeros(n, H, device=DEV)
    abg = ab.to(DEV)
    ops.edgeconv(Mat.of(abg, 0, H), Mat.of(abg, H, H), ops.csr_build(ei.to(DEV), n), packing.to_device(ec, DEV), Mat.of(out))
    torch.cuda.synchronize()
    assert out_ref.max().item() < 0
    assert maxdiff(out, out_ref) <= 2e-5 * out_ref.abs().max().item()


def test_small_ops(ops):
    g = torch.Generator().manual_seed(1)
    emu = EmuOps()
    src = torch.randn(33, 15, generator=g)
    dst_ref, dst = torch.zeros(40, 9), torch.zeros(40, 9, device=DEV)
    emu.copy2d(Mat.of(src, 3, 3), Mat.of(dst_ref, 5, 3, 7, 33))
    ops.copy2d(Mat.of(src.to(DEV), 3, 3), Mat.of(dst, 5, 3, 7, 33))
    assert torch.equal(dst.cpu(), dst_ref)
    cols = torch.tensor([0, 2, 5, 14], dtype=torch.int32)
    o_ref, o = torch.zeros(33, 6), torch.zeros(33, 6, device=DEV)
    emu.gather_cols(Mat.of(src), cols, Mat.of(o_ref, 1, 4))
    ops.gather_cols(Mat.of(src.to(DEV)), cols.to(DEV), Mat.of(o, 1, 4))
    assert torch.equal(o.cpu(), o_ref)
    batch = torch.tensor([0, 0, 1, 1, 1, 2])
    assert torch.equal(ops.make_seg(batch.to(DEV), 3, 2).cpu(), emu.make_seg(batch, 3, 2))
    x = torch.randn(5 * 37, 32, generator=g)
    x[3] = 0.0                                                      # zero row: eps clamp, no NaN
    y_ref, y = torch.zeros(37, 5, 32), torch.zeros(37, 5, 32, device=DEV)
    emu.rownorm(Mat.of(x), 37, 5, y_ref, 160, 32)
    ops.rownorm(Mat.of(x.to(DEV)), 37, 5, y, 160, 32)
    assert maxdiff(y, y_ref) <= 1e-6 and not torch.isnan(y).any()
    for cols_n, rows_n in ((64, 1003), (33, 50), (32, 4099)):      # 16-byte vector kernel (32 / 64 columns) and the general one
        xn = torch.randn(rows_n, cols_n + (4 - cols_n % 4) % 4, generator=g)
        yn_ref, yn = torch.zeros(rows_n, cols_n), torch.zeros(rows_n, cols_n, device=DEV)
        emu.rownorm(Mat.of(xn, 0, cols_n), rows_n, 1, yn_ref, cols_n, 0)
        ops.rownorm(Mat.of(xn.to(DEV), 0, cols_n), rows_n, 1, yn, cols_n, 0)
        assert maxdiff(yn, yn_ref) <= 1e-6
    xa = torch.nn.functional.normalize(torch.randn(70, 5, 32, generator=g), dim=2)
    gq, cls = torch.randn(2, 32, generator=g), torch.randn(32, generator=g)
    a_ref, a = torch.zeros(70, 64), torch.zeros(70, 64, device=DEV)
    emu.cls_attention(xa, gq, cls, Mat.of(a_ref))
    ops.cls_attention(xa.to(DEV), gq.to(DEV), cls.to(DEV), Mat.of(a))
    assert maxdiff(a, a_ref) <= 2e-6
    for mode in ("mean", "max"):
        r_ref, r = torch.zeros(70, 32), torch.zeros(70, 32, device=DEV)
        emu.frame_reduce(xa, mode, Mat.of(r_ref))
        ops.frame_reduce(xa.to(DEV), mode, Mat.of(r))
        assert maxdiff(r, r_ref) <= 1e-6


def test_cpu_tensors_are_refused(ops):
    from morig_amd.native import MorigNativeError
    with pytest.raises(MorigNativeError):
        ops.copy2d(Mat.of(torch.zeros(2, 2)), Mat.of(torch.zeros(2, 2)))


# ---------------------------------------------------------------------------------------------------
# CorrNet point-branch kernels
# ---------------------------------------------------------------------------------------------------
def _clouds(counts, seed):
    g = torch.Generator().manual_seed(seed)
    pos = torch.rand(sum(counts), 3, generator=g)
    pos4 = torch.zeros(sum(counts), 4)
    pos4[:, :3] = pos
    ptr = torch.tensor([0] + list(torch.tensor(counts).cumsum(0)), dtype=torch.int32)
    return pos4, ptr


@pytest.mark.parametrize("counts,ratio,rand", [([700, 33, 1500], 0.5, False), ([8192, 4096], 0.25, True), ([5], 0.5, False)])
def test_fps_bit_exact(ops, counts, ratio, rand):
    import math
    pos4, ptr = _clouds(counts, 7)
    newc = [math.ceil(ratio * c) for c in counts]
    optr = torch.tensor([0] + list(torch.tensor(newc).cumsum(0)), dtype=torch.int32)
    start = torch.tensor([c // 3 for c in counts], dtype=torch.int32) if rand else None
    want = EmuOps().fps(Mat.of(pos4, 0, 3), ptr, optr, start, len(counts), max(counts), sum(newc))
    got = ops.fps(Mat.of(pos4.to(DEV), 0, 3), ptr.to(DEV), optr.to(DEV), None if start is None else start.to(DEV),
                  len(counts), max(counts), sum(newc))
    assert torch.equal(got.cpu(), want)


def test_ball_query_and_bipartite_csr_bit_exact(ops):
    counts = [900, 300]
    pos4, ptr = _clouds(counts, 3)
    cidx = torch.cat([torch.arange(0, 900, 3), 900 + torch.arange(0, 300, 2)])
    cen = pos4[cidx].contiguous()
    cptr = torch.tensor([0, 300, 450], dtype=torch.int32)
    emu = EmuOps()
    for r, mx in ((0.12, 64), (0.3, 16)):
        want = emu.ball_query(Mat.of(pos4, 0, 3), ptr, Mat.of(cen, 0, 3), cptr, 2, r, mx)
        got = ops.ball_query(Mat.of(pos4.to(DEV), 0, 3), ptr.to(DEV), Mat.of(cen.to(DEV), 0, 3), cptr.to(DEV), 2, r, mx)
        assert torch.equal(got.cpu(), want)
        cw = emu.csr_build(want, cen.shape[0], n_src=pos4.shape[0], skip_negative=True)
        cg = ops.csr_build(got, cen.shape[0], n_src=pos4.shape[0], skip_negative=True)
        torch.cuda.synchronize()
        assert int(cg.status.item()) == 0 and torch.equal(cg.rowptr.cpu(), cw.rowptr)
        E = int(cw.rowptr[-1])
        assert _segments(cg, E) == _segments(cw, E)
        cs = ops.csr_from_slots(got, cen.shape[0], mx, pos4.shape[0])          # the atomics-free builder for slot tables
        torch.cuda.synchronize()
        assert int(cs.status.item()) == 0 and torch.equal(cs.rowptr.cpu(), cw.rowptr)
        assert _segments(cs, E) == _segments(cw, E)
        assert torch.equal(cs.dst.cpu()[:E].long(), torch.repeat_interleave(torch.arange(cen.shape[0]), (cw.rowptr[1:] - cw.rowptr[:-1]).long()))


def test_point_kernels_known_answers_ties_and_over_full_balls(ops):
    """The hand-computed known answers of tests/test_oracle_kat.py (VERDICT r3 #4c) on the HIP kernels themselves, so the device
    path is pinned to the published torch_cluster / PyG behaviour directly and not only through the restated oracle:
    over-full ball with coincident points (first 64 hits in index order, strict <), FPS arg-max ties (lowest index), and the
    bipartite self-loop step of PointConv (raw pair (k, k) dropped, (k, k) appended for every target k)."""
    # -- radius: 6 far-ish points first, one at exactly r, then 32 coincident pairs at distance 0.1
    far = [[0.9, 0.0, 0.0]] * 3 + [[0.0, 0.9, 0.0]] * 3
    x = torch.zeros(71, 4)
    x[:, :3] = torch.tensor(far + [[1.0, 0.0, 0.0]] + [[0.1, 0.0, 0.0], [0.0, 0.1, 0.0]] * 32)
    y = torch.zeros(1, 4)
    ptr_x, ptr_y = torch.tensor([0, 71], dtype=torch.int32), torch.tensor([0, 1], dtype=torch.int32)
    coo = ops.ball_query(Mat.of(x.to(DEV), 0, 3), ptr_x.to(DEV), Mat.of(y.to(DEV), 0, 3), ptr_y.to(DEV), 1, 1.0, 64).cpu()
    assert coo[0].tolist() == [0, 1, 2, 3, 4, 5] + list(range(7, 65)) and coo[1].tolist() == [0] * 64
    coo = ops.ball_query(Mat.of(x.to(DEV), 0, 3), ptr_x.to(DEV), Mat.of(y.to(DEV), 0, 3), ptr_y.to(DEV), 1, 1.0, 16).cpu()
    assert coo[0].tolist() == [0, 1, 2, 3, 4, 5] + list(range(7, 17))
    # -- FPS ties: start 0, the extremes -2 / +2 tie -> lower index; then the mid points -1 / +1 tie -> lower index
    for pts, extremes in (([[0.0, 0, 0], [-2.0, 0, 0], [2.0, 0, 0], [1.0, 0, 0], [-1.0, 0, 0], [0.0, 0.5, 0]], [-2.0, 1.0]),
                          ([[0.0, 0, 0], [2.0, 0, 0], [-2.0, 0, 0], [-1.0, 0, 0], [1.0, 0, 0], [0.0, 0.5, 0]], [2.0, -1.0])):
        p4 = torch.zeros(6, 4)
        p4[:, :3] = torch.tensor(pts)
        got = ops.fps(Mat.of(p4.to(DEV), 0, 3), torch.tensor([0, 6], dtype=torch.int32, device=DEV),
                      torch.tensor([0, 4], dtype=torch.int32, device=DEV), None, 1, 6, 4).cpu()
        assert got.tolist() == [0, 1, 2, 3]
        assert [p4[1, 0].item(), p4[3, 0].item()] == extremes
    # -- bipartite self loops, N_src = 5 > N_dst = 2: raw (1, 1) dropped, (0, 0) and (1, 1) appended -> target 0 hears {4, 0},
    #    target 1 hears {3, 1}
    ei = torch.tensor([[4, 1, 3], [0, 1, 1]])
    c = ops.csr_build(ei.to(DEV), 2, n_src=5)
    torch.cuda.synchronize()
    assert int(c.status.item()) == 0 and c.rowptr.cpu().tolist() == [0, 2, 4]
    assert sorted(c.src.cpu()[:2].tolist()) == [0, 4] and sorted(c.src.cpu()[2:4].tolist()) == [1, 3]
    assert c.dst.cpu()[:4].tolist() == [0, 0, 1, 1]
    # fewer sources than targets does not occur on the path (centres are sampled FROM the sources): the C-ABI refuses it
    from morig_amd import native
    with pytest.raises(native.MorigNativeError):
        ops.csr_build(torch.tensor([[1, 0, 1], [0, 2, 2]], device=DEV), 4, n_src=2)


@pytest.mark.parametrize("H,N3", [(32, 64), (64, 128), (256, 256)])
def test_pointconv_two_pass(ops, H, N3):
    g = torch.Generator().manual_seed(H)
    n_src, n_dst = 1200, 500
    src = torch.randint(0, n_src, (9000,), generator=g)
    dst = torch.randint(0, n_dst, (9000,), generator=g)
    src[::5] = -1
    ei = torch.stack([src, torch.where(src < 0, torch.full_like(dst, -1), dst)])
    emu = EmuOps()
    ec = _edge_pack(H, 4)
    lin = _lin(N3, H, 6)
    A, Bm = torch.randn(n_dst, H, generator=g), torch.randn(n_src, H, generator=g)
    cw = emu.csr_build(ei, n_dst, n_src=n_src, skip_negative=True)
    zw = torch.full((cw.capacity, H), float("nan"))
    emu.edge_hidden(Mat.of(A), Mat.of(Bm), cw, ec, Mat.of(zw))
    ow = torch.zeros(n_dst, N3)
    emu.segmax_gemm(Mat.of(zw), lin, True, cw, Mat.of(ow))
    cg = ops.csr_build(ei.to(DEV), n_dst, n_src=n_src, skip_negative=True)
    zg = torch.zeros(cg.capacity, H, device=DEV)
    ops.edge_hidden(Mat.of(A.to(DEV)), Mat.of(Bm.to(DEV)), cg, packing.to_device(ec, DEV), Mat.of(zg))
    og = torch.zeros(n_dst, N3, device=DEV)
    ops.segmax_gemm(Mat.of(zg), packing.to_device(lin, DEV), True, cg, Mat.of(og))
    torch.cuda.synchronize()
    assert maxdiff(og, ow) <= 2e-5 * max(1.0, ow.abs().max().item())


def _pointconv_module(cx, H, H3, seed):
    """a PointConv local_nn = MLP([cx+3, H, H, H3]) with non-trivial BatchNorm statistics (negative gammas included)"""
    g = torch.Generator().manual_seed(seed)
    dims = [cx + 3, H, H, H3]
    layers = []
    for i in range(3):
        lin = torch.nn.Linear(dims[i], dims[i + 1])
        bn = torch.nn.BatchNorm1d(dims[i + 1])
        with torch.no_grad():
            lin.weight.copy_(torch.randn(dims[i + 1], dims[i], generator=g) / dims[i] ** 0.5)
            lin.bias.copy_(torch.randn(dims[i + 1], generator=g) * 0.1)
            bn.weight.copy_(torch.randn(dims[i + 1], generator=g))
            bn.bias.copy_(torch.randn(dims[i + 1], generator=g) * 0.2)
            bn.running_mean.copy_(torch.randn(dims[i + 1], generator=g) * 0.3)
            bn.running_var.copy_(torch.rand(dims[i + 1], generator=g) + 0.5)
        layers.append(torch.nn.Sequential(lin, torch.nn.ReLU(), bn))
    return torch.nn.Sequential(*layers).eval()


@pytest.mark.parametrize("H,H3,cx", [(32, 64, 0), (64, 128, 64)])
@pytest.mark.parametrize("n_src,n_ctr,fill", [(700, 300, "mixed"), (9000, 4100, "full"), (40, 33, "sparse")])
def test_pointconv_fused_matches_the_slot_table_definition_and_the_two_pass_path(ops, H, H3, cx, n_src, n_ctr, fill):
    """morig_pointconv_fused against (a) the fp32 definition written from the slot table (tests/emulate.py) and (b) the HIP
    two-pass path over morig_csr_from_slots: full tables (the self loop has no free slot), tables with unused slots, slots
    naming the centre's own index (dropped by remove_self_loops), centres whose only edge is the self loop."""
    g = torch.Generator().manual_seed(n_src + H)
    nn_ = _pointconv_module(cx, H, H3, 3)
    pk = packing.pack_pointconv(nn_, cx)
    assert pk["fused"] is not None
    slots = torch.randint(0, n_src, (n_ctr, 64), generator=g)
    if fill != "full":
        drop = torch.rand(n_ctr, 64, generator=g) < (0.3 if fill == "mixed" else 0.95)
        slots[drop] = -1
        slots[5] = -1                                           # only the self loop
    slots[7, 3] = 7                                             # source index == centre index: removed
    slots[11] = torch.arange(64) % 3 + 1                        # duplicates of three sources
    coo = torch.stack([slots.reshape(-1), torch.arange(n_ctr).repeat_interleave(64)])
    coo[1][coo[0] < 0] = -1
    A, Bm = torch.randn(n_ctr, H, generator=g), torch.randn(n_src, H, generator=g)
    emu = EmuOps()
    emu.emulate_split = True
    want = torch.zeros(n_ctr, H3)
    emu.pointconv_fused(Mat.of(A), Mat.of(Bm), coo, 64, pk, Mat.of(want))
    pkd = packing.to_device(pk, DEV)
    if not ops.fast:
        assert not ops.pointconv_can_fuse(pkd, 64)              # fp32 MFMA mode: the callers take the two-pass path
        return
    assert ops.pointconv_can_fuse(pkd, 64)
    Ad, Bd, cood = A.to(DEV), Bm.to(DEV), coo.to(DEV)
    got = torch.full((n_ctr, H3 + 4), float("nan"), device=DEV)
    ops.pointconv_fused(Mat.of(Ad), Mat.of(Bd), cood, 64, pkd, Mat.of(got, 0, H3))
    two = torch.zeros(n_ctr, H3, device=DEV)
    csr = ops.csr_from_slots(cood, n_ctr, 64, n_src)
    z = torch.zeros(csr.capacity, H, device=DEV)
    ops.edge_hidden(Mat.of(Ad), Mat.of(Bd), csr, pkd["edge"], Mat.of(z))
    ops.segmax_gemm(Mat.of(z), pkd["last"], True, csr, Mat.of(two))
    torch.cuda.synchronize()
    assert bool(torch.isnan(got[:, H3:]).all()), "wrote outside its window"
    scale = max(1.0, want.abs().max().item())
    assert maxdiff(got[:, :H3], want) <= 2e-5 * scale
    assert maxdiff(got[:, :H3], two.cpu()) <= 2e-5 * scale
    again = torch.zeros(n_ctr, H3, device=DEV)
    ops.pointconv_fused(Mat.of(Ad), Mat.of(Bd), cood, 64, pkd, Mat.of(again))
    assert torch.equal(again, got[:, :H3]), "not deterministic"


def test_pointconv_fused_flags_bad_slots_and_overflow(ops):
    if not ops.fast:
        pytest.skip("split-fp16 kernel")
    nn_ = _pointconv_module(0, 32, 64, 5)
    pkd = packing.to_device(packing.pack_pointconv(nn_, 0), DEV)
    slots = torch.randint(0, 100, (50, 64))
    coo = torch.stack([slots.reshape(-1), torch.arange(50).repeat_interleave(64)]).to(DEV)
    A, Bm = torch.randn(50, 32, device=DEV), torch.randn(100, 32, device=DEV)
    out = torch.zeros(50, 64, device=DEV)
    bad = coo.clone()
    bad[0, 17] = 100                                            # one past the last source row
    with pytest.raises(native.MorigNativeError):
        ops.pointconv_fused(Mat.of(A), Mat.of(Bm), bad, 64, pkd, Mat.of(out))
    flag = ops._flag(A.device)
    flag.zero_()
    ops.pointconv_fused(Mat.of(A), Mat.of(Bm), coo, 64, pkd, Mat.of(out))
    assert int(flag.item()) == 0
    big = Bm.clone()
    big[3, 5] = 7.0e4                                           # outside the fp16 range after the first ReLU
    ops.pointconv_fused(Mat.of(A), Mat.of(big), coo, 64, pkd, Mat.of(out))
    assert int(flag.item()) == 1
    flag.zero_()


@pytest.mark.parametrize("k", [1, 3])
def test_knn_interpolate(ops, k):
    xs, px = _clouds([300, 1, 2500], 5)
    ys, py = _clouds([700, 40, 5000], 6)
    ys[3, :3] = xs[10, :3]                       # coincident point: 1e-16 clamp
    g = torch.Generator().manual_seed(2)
    feat = torch.randn(xs.shape[0], 20, generator=g)
    want = torch.zeros(ys.shape[0], 24)
    EmuOps().knn_interpolate(Mat.of(feat), Mat.of(xs, 0, 3), px, Mat.of(ys, 0, 3), py, 3, 5000, k, Mat.of(want, 0, 20))
    got = torch.zeros(ys.shape[0], 24, device=DEV)
    ops.knn_interpolate(Mat.of(feat.to(DEV)), Mat.of(xs.to(DEV), 0, 3), px.to(DEV), Mat.of(ys.to(DEV), 0, 3), py.to(DEV),
                        3, 5000, k, Mat.of(got, 0, 20))
    assert maxdiff(got, want) <= 1e-5 * max(1.0, want.abs().max().item())


def test_cosine_nn_and_gather_rows(ops):
    g = torch.Generator().manual_seed(8)
    v = torch.nn.functional.normalize(torch.randn(700, 64, generator=g), dim=1)
    p = torch.nn.functional.normalize(torch.randn(1900, 64, generator=g), dim=1)
    pv = torch.tensor([0, 300, 700], dtype=torch.int32)
    pp = torch.tensor([0, 900, 1900], dtype=torch.int32)
    nw, sw = EmuOps().cosine_nn(Mat.of(v), pv, Mat.of(p), pp, 2, 400)
    ng, sg = ops.cosine_nn(Mat.of(v.to(DEV)), pv.to(DEV), Mat.of(p.to(DEV)), pp.to(DEV), 2, 400)
    assert maxdiff(sg, sw) <= 2e-6
    assert (ng.cpu() != nw).float().mean().item() <= 0.005       # only fp-order near-ties may differ
    idx = torch.tensor([5, 0, -1, 1899], dtype=torch.int32)
    out = torch.zeros(4, 70, device=DEV)
    ops.gather_rows(Mat.of(p.to(DEV)), idx.to(DEV), Mat.of(out, 3, 64))
    assert torch.equal(out.cpu()[0, 3:67], p[5]) and float(out[2].abs().sum()) == 0 and torch.equal(out.cpu()[3, 3:67], p[1899])


def test_sigmoid_minmax(ops):
    """deformnet.py:42-46: sigmoid, then per-mesh min-max normalisation (ragged meshes, one single-vertex-pair mesh)."""
    g = torch.Generator().manual_seed(21)
    x = torch.randn(1000, 3, generator=g) * 2.0
    ptr = torch.tensor([0, 2, 300, 1000], dtype=torch.int32)
    want = torch.full((1000, 1), float("nan"))
    EmuOps().sigmoid_minmax(Mat.of(x, 1, 1), ptr, 3, Mat.of(want))
    got = torch.zeros(1000, 2, device=DEV)
    ops.sigmoid_minmax(Mat.of(x.to(DEV), 1, 1), ptr.to(DEV), 3, Mat.of(got, 1, 1))
    assert maxdiff(got[:, 1:2], want) <= 2e-6
    assert float(got[:, 0].abs().sum()) == 0
    for b in range(3):
        seg = got[int(ptr[b]):int(ptr[b + 1]), 1]
        assert float(seg.min()) == 0.0 and float(seg.max()) == 1.0


@pytest.mark.parametrize("k", [1, 5, 8])
def test_cosine_knn_all_rows(ops, k):
    """knn(pts_f, vtx_f, k, cosine=True) (deformnet.py:49): order, lowest index on ties, -1 padding in a cloud with < k candidates."""
    from oracle import pyg_primitives as P
    g = torch.Generator().manual_seed(9)
    y = torch.nn.functional.normalize(torch.randn(700, 64, generator=g), dim=1)
    x = torch.nn.functional.normalize(torch.randn(1203, 64, generator=g), dim=1)
    x[40] = x[7]; x[41] = x[7]; y[3] = x[7]                        # exact ties -> lowest index first
    py = torch.tensor([0, 300, 650, 700], dtype=torch.int32)
    px = torch.tensor([0, 900, 1200, 1203], dtype=torch.int32)    # last cloud: 3 candidates only
    got = ops.cosine_knn(Mat.of(y.to(DEV)), py.to(DEV), Mat.of(x.to(DEV)), px.to(DEV), 3, 350, k).cpu()
    emu = EmuOps().cosine_knn(Mat.of(y), py, Mat.of(x), px, 3, 350, k)
    # primitive of the oracle: [y_idx ; x_idx] pairs, nearest first
    by = torch.repeat_interleave(torch.arange(3), torch.tensor([300, 350, 50]))
    bx = torch.repeat_interleave(torch.arange(3), torch.tensor([900, 300, 3]))
    yi, xi = P.knn(x, y, k, bx, by, cosine=True)
    want = torch.full((700, k), -1, dtype=torch.int32)
    pos = torch.zeros(700, dtype=torch.long)
    for a, b in zip(yi.tolist(), xi.tolist()):
        want[a, pos[a]] = b; pos[a] += 1
    assert torch.equal(emu, want)
    sim = lambda idx: torch.where(idx >= 0, (y[:, None, :] * x[idx.long().clamp(min=0)]).sum(-1), torch.full(idx.shape, -9.0))
    assert maxdiff(sim(got), sim(want)) <= 2e-6                   # same similarity profile
    assert (got != want).float().mean().item() <= 0.005           # only fp-order near-ties may differ
    assert torch.equal(got < 0, want < 0)
    if k >= 3:
        tied = (want == 7).any(1) & (want == 40).any(1) & (want == 41).any(1)
        assert bool(tied[3])
        for r in torch.nonzero(tied).flatten().tolist():
            row = got[r].tolist()
            assert row.index(7) < row.index(40) < row.index(41)


def test_cosine_knn_visible_invisible_split_and_votes(ops):
    """deformnet.py:57-95: rows with mask < 0.5 query rows with mask >= 0.5 of their own mesh; then both votes."""
    g = torch.Generator().manual_seed(10)
    n, k = 900, 5
    f = torch.nn.functional.normalize(torch.randn(n, 64, generator=g), dim=1)
    pf = torch.nn.functional.normalize(torch.randn(1500, 64, generator=g), dim=1)
    vis = torch.rand(n, 1, generator=g)
    ptr = torch.tensor([0, 400, 897, 900], dtype=torch.int32)
    pptr = torch.tensor([0, 700, 1400, 1500], dtype=torch.int32)
    vis[0:400] = (vis[0:400] > 0.3).float() * 0.8 + 0.1            # plenty of both kinds
    vis[897:900] = torch.tensor([[0.9], [0.2], [0.1]])             # mesh 2: ONE visible vertex -> 1 neighbour, -1 padding
    pos = torch.randn(n, 3, generator=g); ppos = torch.randn(1500, 3, generator=g)
    e = EmuOps()
    want_idx = e.cosine_knn(Mat.of(f), ptr, Mat.of(f), ptr, 3, 500, k, vis=Mat.of(vis), split=True)
    fd, vd = f.to(DEV), vis.to(DEV)
    got_idx = ops.cosine_knn(Mat.of(fd), ptr.to(DEV), Mat.of(fd), ptr.to(DEV), 3, 500, k, vis=Mat.of(vd), split=True).cpu()
    assert torch.equal(got_idx < 0, want_idx < 0)
    assert (got_idx != want_idx).float().mean().item() <= 0.005
    assert bool((got_idx[(vis >= 0.5).squeeze(1)] == -1).all())
    assert got_idx[898].tolist() == [897, -1, -1, -1, -1]
    sel = got_idx[got_idx >= 0].long()
    assert bool((vis[sel] >= 0.5).all())
    # votes (same neighbour lists on both sides)
    idx1 = e.cosine_knn(Mat.of(f), ptr, Mat.of(pf), pptr, 3, 500, k)
    want = torch.full((n, 4), float("nan"))
    e.flow_vote(0, idx1, Mat.of(f), Mat.of(pf), Mat.of(pos), Mat.of(ppos), Mat.of(vis), Mat.of(want))
    e.flow_vote(1, want_idx, Mat.of(f), Mat.of(f), None, None, Mat.of(vis), Mat.of(want))
    got = torch.full((n, 6), 7.0, device=DEV)
    l1 = Mat.of(got, 1, 4)
    ops.flow_vote(0, idx1.to(DEV), Mat.of(fd), Mat.of(pf.to(DEV)), Mat.of(pos.to(DEV)), Mat.of(ppos.to(DEV)), Mat.of(vd), l1)
    ops.flow_vote(1, want_idx.to(DEV), Mat.of(fd), Mat.of(fd), None, None, Mat.of(vd), l1)
    assert maxdiff(got[:, 1:5], want) <= 1e-4 * max(1.0, want.abs().max().item())
    assert bool((got[:, 0] == 7.0).all()) and bool((got[:, 5] == 7.0).all())


def test_split_fp16_overflow_is_flagged_and_forward_falls_back():
    """operands beyond the fp16 range: the fast kernels raise the flag, NativeOps.guarded re-runs in fp32."""
    from morig_amd import native, synth
    from morig_amd.models import basic_modules as bm
    from oracle import nets
    o = native.get_ops()
    o.precision = "f16x3"
    g = torch.Generator().manual_seed(0)
    x = torch.randn(300, 8, generator=g) * 3.0e5                  # way outside fp16
    lin = _lin(64, 8, 3, bn=False)
    flag = o._flag(torch.device("cuda", torch.cuda.current_device()))
    flag.zero_()
    y = torch.zeros(300, 64, device=DEV)
    o.gemm(Mat.of(x.to(DEV)), packing.to_device(lin, DEV), False, Y=Mat.of(y))
    assert int(flag.item()) == 1
    # module level: huge features through a GCU still match the oracle (fp32 redo)
    batch = synth.make_batch([5], n_side=10)
    feat = batch.pos * 1.0e6
    ours = synth.load_recipe(bm.GCU(3, 32).eval(), 9).to(DEV)
    ref = synth.load_recipe(nets.GraphConvUnit(3, 32).eval(), 9)
    want = ref(feat, batch.tpl_edge_index, batch.geo_edge_index)
    got = ours(feat.to(DEV), batch.tpl_edge_index.to(DEV), batch.geo_edge_index.to(DEV))
    assert torch.isfinite(got).all()
    assert maxdiff(got, want) <= 1e-4 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("M,N,K", [(4096, 512, 544), (1100, 256, 288), (1024, 256, 64), (33000, 512, 96), (2049, 768, 320)])
@pytest.mark.parametrize("y_split,relu,bn", [(False, True, True), (True, True, True), (False, False, False)])
def test_gemm_fp32_x_wide_outputs(M, N, K, y_split, relu, bn):
    """fp32 X, N a multiple of 256, K a multiple of 32 (the GCU vertex MLP shapes K = 544 -> 512 and 288 -> 256 among them): ragged
    last row tile, fp32 and split-layout outputs, plain Linear. (Written for the producer / consumer measurement kernel of
    DESIGN.md section 5 [r03]; kept for the tile engine, which runs these shapes.)"""
    from morig_amd import native
    o = native.get_ops()
    o.precision = "f16x3"
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K + 4, generator=g) * 3.0
    x[:, K:] = float("nan")                                  # nothing beyond K may be read into the sums
    lin = _lin(N, K, 7, bn)
    want = torch.zeros(M, N)
    EmuOps().gemm(Mat.of(x, 0, K), lin, relu, Y=Mat.of(want))
    xg, ling = x.to(DEV), packing.to_device(lin, DEV)
    if y_split:
        ys = torch.zeros(M, N + 32, device=DEV)
        o.gemm(Mat.of(xg, 0, K), ling, relu, Y=Mat.of(ys, 32, N), y_split=True)
        torch.cuda.synchronize()
        got = packing.unsplit_f16(ys.cpu(), N + 32)[:, 32:32 + N]
        assert float(ys[:, :32].abs().sum()) == 0
    else:
        yb = torch.zeros(M, N + 8, device=DEV)
        o.gemm(Mat.of(xg, 0, K), ling, relu, Y=Mat.of(yb, 4, N))
        torch.cuda.synchronize()
        got = yb[:, 4:4 + N].cpu()
        assert float(yb[:, :4].abs().sum()) == 0 and float(yb[:, 4 + N:].abs().sum()) == 0
    assert maxdiff(got, want) <= 2e-5 * max(1.0, want.abs().max().item())


# (launches of at most 256 tiles of 256 x 256 run on the 128 x 128 LDS-DMA kernel [r06], larger ones on the 256 x 256 kernels: both sizes)
@pytest.mark.parametrize("M,N,K", [(300, 512, 544), (1000, 1024, 867), (129, 64, 64), (4096, 256, 1024), (17000, 1024, 867), (66000, 256, 96)])
def test_gemm_split_fp16_activation_layout(M, N, K):
    """X consumed and Y produced in the split-fp16 activation layout (chunk = [32 hi | 32 lo])."""
    from morig_amd import native
    o = native.get_ops()
    o.precision = "f16x3"
    g = torch.Generator().manual_seed(M + K)
    Kp, Np = (K + 31) // 32 * 32, (N + 31) // 32 * 32
    x = torch.zeros(M, Kp + 32)
    x[:, :K] = torch.randn(M, K, generator=g)
    lin = _lin(N, K, 4)
    want = torch.zeros(M, N)
    EmuOps().gemm(Mat.of(x, 0, K), lin, True, Y=Mat.of(want))
    xs = packing.split_f16(x).to(DEV)                           # same function that pre-splits the weights
    ys = torch.zeros(M, Np + 32, device=DEV)
    o.gemm(Mat.of(xs, 0, K), packing.to_device(lin, DEV), True, Y=Mat.of(ys, 32, N), x_split=True, y_split=True)
    torch.cuda.synchronize()
    got = packing.unsplit_f16(ys.cpu(), Np + 32)[:, 32:32 + N]
    assert maxdiff(got, want) <= 2e-5 * max(1.0, want.abs().max().item())
    # x fp32 -> y split, and x split -> y fp32
    y2 = torch.zeros(M, N, device=DEV)
    o.gemm(Mat.of(xs, 0, K), packing.to_device(lin, DEV), True, Y=Mat.of(y2), x_split=True)
    assert maxdiff(y2, want) <= 2e-5 * max(1.0, want.abs().max().item())
    # pooled consumer of a split X
    seg = torch.sort(torch.randint(0, 5, (M,), generator=g))[0].int()
    pw = torch.zeros(5, N)
    EmuOps().gemm(Mat.of(x, 0, K), lin, True, seg=seg, pool=pw)
    pg = torch.zeros(5, N, device=DEV)
    o.gemm(Mat.of(xs, 0, K), packing.to_device(lin, DEV), True, seg=seg.to(DEV), pool=pg, x_split=True)
    present = torch.unique(seg.long())
    assert maxdiff(pg[present.to(DEV)], pw[present]) <= 2e-5 * max(1.0, pw[present].abs().max().item())


@pytest.mark.parametrize("M,N,K", [(1000, 1024, 867), (700, 512, 256), (4133, 256, 1024), (513, 1024, 64), (2600, 1024, 899),
                                   (17000, 1024, 800), (16500, 1024, 288), (33000, 512, 64), (66000, 256, 1024)])
@pytest.mark.parametrize("segs", ["one", "tiles", "ragged"])
@pytest.mark.parametrize("y_split,relu", [(False, False), (True, True)])
def test_gemm_dma_register_epilogue(M, N, K, segs, y_split, relu):
    """the LDS-DMA GEMMs' store launches (gemm_dma.hip one tile per workgroup, gemm_dmap.hip persistent; the first five shapes are
    few-tile launches, which run on the 128 x 128 kernel [r06], the last four fill more than 256 tiles of 256 x 256) run their MFMAs
    transposed and store from registers (epilogue_store.h: store_tile_regs): bias and -- when a 256-row tile lies in one mesh --
    the row bias are the accumulators' initial value ("one", "tiles": every tile in one mesh), otherwise the row bias is added
    per row ("ragged": mesh boundaries inside tiles); fp32 and split-fp16 outputs, rows past M never written."""
    from morig_amd import native
    o = native.get_ops()
    o.precision = "f16x3"
    g = torch.Generator().manual_seed(M + K + len(segs))
    Kp, Np = (K + 31) // 32 * 32, (N + 31) // 32 * 32
    x = torch.zeros(M, Kp)
    x[:, :K] = torch.randn(M, K, generator=g)
    lin = _lin(N, K, 6, bn=relu)
    if segs == "one":
        seg = torch.zeros(M, dtype=torch.int32)
    elif segs == "tiles":
        seg = (torch.arange(M) // 512).int()
    else:
        seg = torch.sort(torch.randint(0, 9, (M,), generator=g))[0].int()
    nseg = int(seg.max()) + 1
    rb = torch.randn(nseg, N, generator=g)
    xs = packing.split_f16(x).to(DEV)
    for use_rb in (False, True):
        want = torch.zeros(M, N)
        kw = dict(rowbias=Mat.of(rb), seg=seg) if use_rb else {}
        EmuOps().gemm(Mat.of(x, 0, K), lin, relu, Y=Mat.of(want), **kw)
        kwg = dict(rowbias=Mat.of(rb.to(DEV)), seg=seg.to(DEV)) if use_rb else {}
        if y_split:
            ys = torch.full((M + 3, Np + 32), 5.0, device=DEV)
            o.gemm(Mat.of(xs, 0, K), packing.to_device(lin, DEV), relu, Y=Mat.of(ys, 32, N, 0, M), x_split=True, y_split=True, **kwg)
            torch.cuda.synchronize()
            full = ys.cpu()
            got = packing.unsplit_f16(full[:M].contiguous(), Np + 32)[:, 32:32 + N]
            assert float((full[M:] - 5.0).abs().sum()) == 0 and float((full[:M, :32] - 5.0).abs().sum()) == 0     # window respected
        else:
            yb = torch.full((M + 3, N + 8), 5.0, device=DEV)
            o.gemm(Mat.of(xs, 0, K), packing.to_device(lin, DEV), relu, Y=Mat.of(yb, 4, N, 0, M), x_split=True, **kwg)
            torch.cuda.synchronize()
            full = yb.cpu()
            got = full[:M, 4:4 + N]
            assert float((full[M:] - 5.0).abs().sum()) == 0 and float((full[:M, :4] - 5.0).abs().sum()) == 0 and float((full[:M, 4 + N:] - 5.0).abs().sum()) == 0
        assert maxdiff(got, want) <= 2e-5 * max(1.0, want.abs().max().item()), (use_rb,)


@pytest.mark.parametrize("n,R,N,K,tc", [(517, 5, 512, 544, 32), (300, 1, 256, 288, 32), (250, 4, 256, 96, 32), (200, 3, 256, 320, 64),
                                        (4096, 5, 512, 544, 32)])
@pytest.mark.parametrize("y_split", [False, True])
def test_gemm_k_tail_reads_the_replica_invariant_block_from_one_copy(n, R, N, K, tc, y_split):
    """morig_gemm_args.X_tail [ABI 3]: the last tail_cols input columns of row m come from row (m % tail_rows) of a separate split-layout
    matrix -- the [pos_tpl | pos_geo] block of a GCUMotion unit under the keyframe loop (models/basic_modules.py:216-217 inside
    models/rignet.py:85-86) enters the unit's MLP without being copied into the 5 replicas' rows. Against the same GEMM on the
    materialised concatenation (torch, CPU); tiles that straddle a replica boundary (n % 256 != 0) included."""
    o = native.get_ops()
    o.precision = "f16x3"
    g = torch.Generator().manual_seed(n + K)
    M, Km = n * R, K - tc
    xm = torch.randn(M, Km, generator=g)
    xt = torch.randn(n, tc, generator=g)
    lin = _lin(N, K, 8, bn=True)
    full = torch.cat([xm, xt.repeat(R, 1)], dim=1).contiguous()
    want = torch.zeros(M, N)
    EmuOps().gemm(Mat.of(full), lin, True, Y=Mat.of(want))
    ling = packing.to_device(lin, DEV)
    xs = packing.split_f16(xm.contiguous()).to(DEV)
    ts = packing.split_f16(xt.contiguous()).to(DEV)
    Np = N
    if y_split:
        ys = torch.full((M + 3, Np + 32), 5.0, device=DEV)
        Y = Mat.of(ys, 32, N, 0, M)
    else:
        ys = torch.full((M + 3, N + 8), 5.0, device=DEV)
        Y = Mat.of(ys, 4, N, 0, M)
    assert o.gemm_takes_tail(ling, Y, tc)
    o.gemm(Mat.of(xs), ling, True, Y=Y, x_split=True, y_split=y_split, x_tail=Mat.of(ts))
    torch.cuda.synchronize()
    fullg = ys.cpu()
    if y_split:
        got = packing.unsplit_f16(fullg[:M].contiguous(), Np + 32)[:, 32:32 + N]
        assert float((fullg[M:] - 5.0).abs().sum()) == 0 and float((fullg[:M, :32] - 5.0).abs().sum()) == 0
    else:
        got = fullg[:M, 4:4 + N]
        assert float((fullg[M:] - 5.0).abs().sum()) == 0 and float((fullg[:M, :4] - 5.0).abs().sum()) == 0
    assert maxdiff(got, want) <= 2e-5 * max(1.0, want.abs().max().item())


def test_gemm_k_tail_is_refused_where_the_kernel_has_none():
    """narrow outputs / pooled launches have no K tail: MORIG_E_UNSUPPORTED (the plans ask ``gemm_takes_tail`` first), bad tail
    geometry is MORIG_E_INVALID"""
    o = native.get_ops()
    o.precision = "f16x3"
    lin = packing.to_device(_lin(64, 96, 3, True), DEV)
    xs = packing.split_f16(torch.randn(256, 64)).to(DEV)
    ts = packing.split_f16(torch.randn(64, 32)).to(DEV)
    y = torch.zeros(256, 64, device=DEV)
    assert not o.gemm_takes_tail(lin, Mat.of(y), 32)
    with pytest.raises(native.MorigNativeError, match="unsupported"):
        o.gemm(Mat.of(xs), lin, True, Y=Mat.of(y), x_split=True, x_tail=Mat.of(ts))
    lin2 = packing.to_device(_lin(256, 96, 3, True), DEV)
    y2 = torch.zeros(256, 256, device=DEV)
    with pytest.raises(native.MorigNativeError, match="invalid"):
        o.gemm(Mat.of(xs), lin2, True, Y=Mat.of(y2), x_split=True, x_tail=Mat.of(torch.zeros(64, 64, device=DEV), 8, 32))   # chunk not 128-byte aligned
    torch.cuda.synchronize()


def test_pack_tails_builds_every_units_chunk_in_one_launch():
    """morig_pack_tails: tail t, row v = [src[v, a_t : a_t + 16] | src[v, b_t : b_t + 16]] as one split-fp16 chunk"""
    o = native.get_ops()
    o.precision = "f16x3"
    g = torch.Generator().manual_seed(5)
    src = torch.randn(1000, 192, generator=g)
    ca, cb = [0, 16, 64, 80, 128, 144], [32, 48, 96, 112, 160, 176]
    got = o.pack_tails(src.to(DEV), ca, cb, 16, 16)
    torch.cuda.synchronize()
    assert got.shape == (6, 1000, 32)
    for t in range(6):
        want = torch.cat([src[:, ca[t]:ca[t] + 16], src[:, cb[t]:cb[t] + 16]], 1)
        back = packing.unsplit_f16(got[t].cpu().contiguous(), 32)
        assert (back - want).abs().max().item() <= 1e-6 * max(1.0, want.abs().max().item())


def test_copy2d_pad_plain_and_split():
    from morig_amd import native
    o = native.get_ops()
    o.precision = "f16x3"
    src = torch.randn(77, 15)
    dst = torch.full((80, 96), 7.0, device=DEV)
    o.copy2d_pad(Mat.of(src.to(DEV), 3, 5), Mat.of(dst, 32, 32, 2, 77))
    d = dst.cpu()
    assert torch.equal(d[2:79, 32:37], src[:, 3:8]) and float(d[2:79, 37:64].abs().sum()) == 0
    assert float((d[:, :32] - 7).abs().sum()) == 0 and float((d[:, 64:] - 7).abs().sum()) == 0
    dst2 = torch.zeros(77, 96, device=DEV)
    o.copy2d_pad(Mat.of(src.to(DEV), 3, 5), Mat.of(dst2, 32, 64), split=True)
    dec = packing.unsplit_f16(dst2.cpu(), 96)
    assert maxdiff(dec[:, 32:37], src[:, 3:8]) <= 1e-6 and float(dec[:, 37:].abs().sum()) == 0


@pytest.mark.parametrize("split", [False, True])
def test_copy2d_rep(ops, split):
    """one launch = R copies: keyframe features (column step 3) and plain replication (column step 0)"""
    if split and not ops.fast:
        pytest.skip("split-fp16 activations only exist on the split-fp16 path")
    g = torch.Generator().manual_seed(31)
    n, R = 500, 5
    flow = torch.randn(n, 16, generator=g)
    e = EmuOps()
    for step, cols in ((3, 3), (0, 4)):
        want = torch.full((R * n + 7, 96), float("nan"))
        got = torch.full((R * n + 7, 96), float("nan"), device=DEV)
        e.copy2d_rep(Mat.of(flow, 1, cols), Mat.of(want, 32, 32, 0, n), R, n, src_col_step=step, split=split)
        ops.copy2d_rep(Mat.of(flow.to(DEV), 1, cols), Mat.of(got, 32, 32, 0, n), R, n, src_col_step=step, split=split)
        if not split:
            assert torch.equal(torch.nan_to_num(got.cpu(), nan=-7.0), torch.nan_to_num(want, nan=-7.0))
        else:
            # split layout: the 32-column chunk holds [32 hi halves | 32 lo halves]; hi + lo reproduces the values
            raw = got.cpu()[: R * n, 32:64].contiguous().view(torch.float16).float().view(R * n, 64)
            assert maxdiff(raw[:, :32] + raw[:, 32:], want[: R * n, 32:64]) <= 1e-6
            assert bool(torch.isnan(got.cpu()[R * n:, :]).all()) and bool(torch.isnan(got.cpu()[:, :32]).all())


def test_fps_full_size_properties(ops):
    """BASELINE.json configs[3] sizes (8192-point clouds, ratio 0.5): the oracle loop is too slow here, so check what
    defines the result: (a) distinct indices inside the cloud, (b) every sample is a farthest point of its step -- its
    distance to the earlier samples equals the maximum over the cloud -- and (c) the lowest index among those maxima."""
    import math
    counts = [8192, 8192, 5000]
    pos4, ptr = _clouds(counts, 11)
    newc = [math.ceil(0.5 * c) for c in counts]
    optr = torch.tensor([0] + list(torch.tensor(newc).cumsum(0)), dtype=torch.int32)
    got = ops.fps(Mat.of(pos4.to(DEV), 0, 3), ptr.to(DEV), optr.to(DEV), None, len(counts), max(counts), sum(newc)).cpu().long()
    for b, (n, m) in enumerate(zip(counts, newc)):
        p0 = int(ptr[b])
        sel = got[int(optr[b]):int(optr[b + 1])] - p0
        assert sel.min() >= 0 and sel.max() < n and len(set(sel.tolist())) == m and int(sel[0]) == 0
        p = pos4[p0:p0 + n, :3].to(DEV)
        dist = torch.full((n,), float("inf"), device=DEV)
        for s in range(1, min(m, 600)):                            # the first 600 steps on the device, same fp32 formula
            d = p - p[sel[s - 1]]
            d = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
            dist = torch.minimum(dist, d)
            mx = dist.max()
            assert float(dist[sel[s]]) == float(mx)
            assert int(torch.nonzero(dist == mx)[0]) == int(sel[s])


def test_cosine_knn_full_size_properties(ops):
    """4096 queries x 8192 candidates per cloud (the benchmark's pair size), k = 5: neighbours come from the query's own
    cloud, similarities are sorted, and nothing outside the list beats its last entry (checked against a torch matmul on
    the device for a sample of rows)."""
    g = torch.Generator().manual_seed(12)
    B, V, P, k = 3, 4096, 8192, 5
    y = torch.nn.functional.normalize(torch.randn(B * V, 64, generator=g), dim=1).to(DEV)
    x = torch.nn.functional.normalize(torch.randn(B * P, 64, generator=g), dim=1).to(DEV)
    py = torch.arange(0, B * V + 1, V, dtype=torch.int32).to(DEV)
    px = torch.arange(0, B * P + 1, P, dtype=torch.int32).to(DEV)
    idx = ops.cosine_knn(Mat.of(y), py, Mat.of(x), px, B, V, k).long()
    assert bool((idx >= 0).all())
    cloud_q = torch.arange(B * V, device=DEV) // V
    assert bool(((idx // P) == cloud_q[:, None]).all())
    rows = torch.randint(0, B * V, (512,), generator=g).to(DEV)
    sims = y[rows] @ x.t()                                                       # [512, B*P]
    own = (torch.arange(B * P, device=DEV)[None, :] // P) == cloud_q[rows][:, None]
    sims = sims.masked_fill(~own, -2.0)
    got = torch.gather(sims, 1, idx[rows])
    assert bool((got[:, :-1] >= got[:, 1:] - 1e-6).all())                        # most similar first
    top = sims.topk(k, dim=1).values
    assert float((got - top).abs().max()) <= 2e-6                               # the same k best similarities


def test_radius_cpu_against_reference_fixture(ops):
    """the inclusive, batch-less ball query of the reference's no-CUDA branch with a uniform random subset
    for over-full rows (morig_radius_sample) -- exact where the reference is deterministic, distributional properties where it
    draws; plus the uniformity of the reservoir itself."""
    from helpers import check_radius_cpu
    check_radius_cpu(DEV)
    # uniformity: one centre, 40 hits, cap 8, many seeds -> every hit kept with probability 8/40 (binomial 3-sigma band)
    x = torch.zeros(40, 4, device=DEV); x[:, 0] = torch.linspace(0, 0.01, 40, device=DEV)
    y = torch.zeros(1, 4, device=DEV)
    hits = torch.zeros(40)
    trials = 4000
    for s in range(trials):
        coo, cnt = ops.radius_sample(Mat.of(x, 0, 3), Mat.of(y, 0, 3), 0.5, 8, 1000 + s)
        hits[coo[0, :8].cpu()] += 1
        assert int(cnt.item()) == 40
    p = 8 / 40
    sigma = (trials * p * (1 - p)) ** 0.5
    assert float((hits - trials * p).abs().max()) < 4.5 * sigma, hits


def test_pooled_gemm_empty_segment_is_zero(ops):
    """a graph id without vertices must pool to 0 (torch_scatter's fill), not to the atomic-max identity (NaN)."""
    g = torch.Generator().manual_seed(3)
    M, K, N, nseg = 500, 64, 128, 4
    x = torch.randn(M, K, generator=g)
    lin = packing.pack_linear(torch.randn(N, K, generator=g) / 8, torch.randn(N, generator=g))
    seg = torch.cat([torch.zeros(200), torch.full((300,), 2.0)]).to(torch.int32)          # ids 1 and 3 are empty
    pool = torch.full((nseg, N), 7.0, device=DEV)
    ops.gemm(Mat.of(x.to(DEV), 0, K), packing.to_device(lin, DEV), True, seg=seg.to(DEV), pool=pool)
    torch.cuda.synchronize()
    assert not torch.isnan(pool).any()
    assert float(pool[1].abs().max()) == 0.0 and float(pool[3].abs().max()) == 0.0 and float(pool[0].abs().max()) > 0


def test_concurrent_forwards_from_two_threads_keep_their_own_guard_state():
    """guard state (overflow flag, forced-fp32 switch, nesting depth) is per thread: two threads running forwards on
    their own streams get the same bits as the sequential runs."""
    import threading
    from morig_amd import models, synth
    mesh = synth.collate([synth.make_mesh(8, n_side=12)]).to(DEV)
    m = synth.load_recipe(models.jointnet_motion(num_keyframes=5, chn_output=3, aggr_method="attn").eval(), 4, mild=True).to(DEV)
    want = m(mesh, mesh.pred_flow)[2].clone()
    outs, errs = {}, []

    def work(i):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st), torch.no_grad():
                for _ in range(3):
                    o = m(mesh, mesh.pred_flow)[2]
                st.synchronize()
            outs[i] = o
        except Exception as e:              # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in ts]; [t.join() for t in ts]
    assert not errs, errs
    assert all(torch.equal(outs[i], want) for i in range(2))
