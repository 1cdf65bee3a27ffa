"""Train-mode forward AND backward of the rignet family on the MI355X-native op layer.

The two starred blocks of the reference -- ``Seq(Linear, ReLU, BatchNorm1d)`` over vertices and the per-edge MLP of
EdgeConvMotion with BatchNorm statistics over EDGES and max aggregation (/root/reference/models/basic_modules.py:31-36,
179-202) -- are ``torch.autograd.Function``s here: ``forward`` runs the native train-mode operators of
``train_forward.py`` and keeps what the backward needs (the ReLU outputs, the batch statistics, the arg-max table of the
aggregation), ``backward`` runs the native backward operators of csrc/train_bwd.hip:

    dense block    dz -> [bn_backward_stats] dgamma, dbeta -> [bn_relu_backward] du -> dX = du W   (morig_gemm on W^T)
                                                                                     dW = du^T X (morig_gemm_tn), db = col sum
    edge MLP       dOut[v] -> [segmax_bn_backward_stats / segmax_bn_relu_backward] du2[e] (dense over edges: BatchNorm2's mean
                   terms reach every edge; the one-hot arg-max gradient itself is never materialised)
                   -> dW2 = du2^T (s1 Z1 + t1), dZ1aff = du2 W2 -> BatchNorm1 + ReLU over edges -> dG[e]
                   -> [edge_scatter_backward] dA[dst], dB[src] -> vertex GEMM backward (dX, dW1 mapped back to [W_a | W_b])

Everything that is not a starred kernel -- concatenations, F.normalize, the 5-token attention, mesh pooling's index
bookkeeping -- stays torch autograd on device tensors, exactly as the reference has it; losses stay in PyTorch
(models/customized_losses.py). Gradient contractions run on the exact-fp32 MFMA kernels: gradients live many orders of
magnitude below activations, outside what the split-fp16 operand format resolves.

``motion_head_step(model, data, input_flow)`` is ``JointNetMotion`` / ``MaskNetMotion.forward`` in training mode
(/root/reference/models/rignet.py:70-133; training/train_rig.py:136-195) with a graph attached: call ``.backward()`` on a
loss of its outputs and every parameter's ``.grad`` is filled, BatchNorm running buffers move as in the reference.
"""
from __future__ import annotations

import contextlib
import math
import os
import weakref
from typing import Optional

import torch
import torch.nn.functional as F

from . import packing
from .native import CSR, Mat
from .runtime import get_ops
from .train_forward import _bn_train, _ld4, _pad_to, batch_moments, set_batchnorm_sync, sync_backward_sums  # noqa: F401


# Arithmetic of the FORWARD contractions in training mode. Default: the exact-fp32 MFMA kernels. The split-fp16 path carries
# ~2 bits less than fp32 per product (2^-22 vs 2^-24); harmless in eval mode (1e-6 of output scale), but a train-mode forward
# of these networks is ill-conditioned (batch statistics over edges feeding arg-max aggregations, six GCNRig passes deep):
# measured on JointNetMotion, outputs and gradients deviate from a float64 run 4-10x more than torch's own float32 run does
# (tests/test_gpu_backward.py). MORIG_TRAIN_PRECISION=f16x3 opts into the fast path (5x the MFMA rate).
def train_fast() -> bool:
    return os.environ.get("MORIG_TRAIN_PRECISION", "f32") == "f16x3"


def _pack_fwd(weight: torch.Tensor, bias: Optional[torch.Tensor], dev):
    return packing.to_device(packing.pack_linear(weight.detach(), None if bias is None else bias.detach(), split=train_fast()), dev)


def _buf(rows: int, cols: int, dev) -> torch.Tensor:
    """[rows, ld4(cols)] fp32 work buffer: uninitialised when there are no padding columns (rows past a CSR's live edge count are
    never read by the row-counted operators, and zero-filling an edges x H buffer is a full HBM pass), zero-filled otherwise"""
    ld = _ld4(cols)
    return torch.empty((rows, ld), dtype=torch.float32, device=dev) if ld == cols else torch.zeros((rows, ld), dtype=torch.float32, device=dev)


def _rows16(x: torch.Tensor) -> torch.Tensor:
    """x as an fp32 matrix whose rows are 16-byte aligned, for ``Mat.of`` (which takes the row pitch from the stride): x itself when it
    already is one -- also as a column window of a wider tensor, which is what the backward of ``torch.cat`` hands out: copying those
    was 2 ms of the step -- otherwise a contiguous copy with the columns zero-padded to a multiple of 4"""
    x = x.detach().float()
    c = x.shape[1]
    if c % 4 == 0 and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.stride(0) >= c and x.data_ptr() % 16 == 0:
        return x
    if c % 4:
        x = F.pad(x, (0, _ld4(c) - c))
    return x.contiguous()


def _pack_f32(weight: torch.Tensor, bias: Optional[torch.Tensor], dev):
    """a packed Linear for the exact-fp32 MFMA kernels (no split image: morig_gemm then takes the fp32 path)"""
    return packing.to_device(packing.pack_linear(weight.detach().float(), None if bias is None else bias.detach().float(), split=False), dev)


def bwd_split() -> bool:
    """MORIG_TRAIN_BWD=f32: the gradient contractions (dX = dU W, dW = dU^T X) on the exact-float32 MFMA kernels; default: the
    bf16 x 3 split (bf16 keeps float32's exponent range, which gradients need; ~16 mantissa bits per operand, 16/3 of the fp32-MFMA rate)"""
    return not os.environ.get("MORIG_TRAIN_BWD", "bf16x3").startswith("f")


def edge_sums_by_pass() -> bool:
    """MORIG_TRAIN_EDGE_SUMS=pass: the first edge layer's BatchNorm sums from a pass over dh and Z1 (fp64 accumulation of float32
    products) instead of the H x H algebra on M = du2^T Z1 (whose entries carry the weight-gradient GEMM's arithmetic: bf16 x 3 by default)"""
    return os.environ.get("MORIG_TRAIN_EDGE_SUMS", "products") == "pass"


def _pack_bwd(weight: torch.Tensor, dev):
    """a packed Linear for a gradient GEMM dX = dU W^T-form: fp32 image always, plus the split-bf16 image unless MORIG_TRAIN_BWD=f32"""
    pk = packing.pack_linear(weight.detach().float(), None, split=False)
    if bwd_split():
        pk.Wsplit_bf16 = packing.split_bf16(pk.W)
    return packing.to_device(pk, dev)


def _gemm_f32(ops, X: Mat, weight, n_out: int) -> torch.Tensor:
    """X @ weight^T on the fp32 MFMA path -> [rows, ld4(n_out)] (columns >= n_out are padding); ``weight``: a tensor, or an
    already packed Linear (``_pack_f32``)"""
    dev = X.base.device
    out = _buf(X.rows, n_out, dev)
    pk = weight if isinstance(weight, packing.PackedLinear) else _pack_f32(weight, None, dev)
    ops.gemm(X, pk, relu=False, Y=Mat.of(out, 0, n_out))
    return out


# Kernel-layout images of the weights, per nn.Module and valid while the parameters they were made from are unchanged: packing a
# Linear is ~10 small launches (zero-padded copy, padded bias / scale / shift vectors), a training step packs ~230 of them forwards
# and again, transposed, backwards -- and motionNet's layers run five times per step on the same weights (measured: ~4 000 of the
# step's 9 300 launches). Keyed on (data_ptr, _version) of the parameters: optimizer.step() updates in place and bumps the version,
# load_state_dict() copies in place likewise. (Edits through ``.data`` bypass the counter: call ``clear_pack_cache()`` after them.)
_PACKS = weakref.WeakKeyDictionary()


class _Packs:
    __slots__ = ("d", "__weakref__")

    def __init__(self):
        self.d = {}

    def get(self, key, params, build):
        ver = tuple((q.data_ptr(), q._version, q.device) for q in params if q is not None) + (train_fast(), bwd_split())
        if _PACK_DEBUG:
            # MORIG_TRAIN_PACK_DEBUG=1: the (storage, version) key cannot see a write through `.data` (an EMA copy, `p.data.clamp_()`):
            # those leave _version alone. The debug key adds a checksum of the values (one device reduction + host read per lookup --
            # slow, for hunting a stale pack); without it call clear_pack_cache() after any such write.
            ver = ver + tuple(float(q.detach().double().sum().item()) + float(q.detach().double().abs().sum().item()) for q in params if q is not None)
        hit = self.d.get(key)
        if hit is None or hit[0] != ver:
            hit = self.d[key] = (ver, build())
        return hit[1]


_PACK_DEBUG = os.environ.get("MORIG_TRAIN_PACK_DEBUG", "0") == "1"


class _NoPacks:
    @staticmethod
    def get(key, params, build):
        return build()


def packs_of(module):
    """the pack cache of one nn.Module (a Linear, or a Seq(Linear, ReLU, BN) layer); MORIG_TRAIN_PACK_CACHE=0 switches it off"""
    if os.environ.get("MORIG_TRAIN_PACK_CACHE", "1") == "0":
        return _NoPacks
    c = _PACKS.get(module)
    if c is None:
        c = _PACKS[module] = _Packs()
    return c


def clear_pack_cache():
    """Drop every cached kernel-layout weight image. The cache follows optimizer steps on its own (in-place updates bump the
    parameters' version counters); call this after writing parameters THROUGH `.data` (EMA copies, clamps, manual loads), which
    no version counter records -- or run with MORIG_TRAIN_PACK_DEBUG=1 to have every lookup verify a checksum."""
    _PACKS.clear()


class _KeyedPacks:
    """a pack cache entry group whose validity follows given LEAF parameters: for blocks whose weight operands are built per call
    (``torch.cat`` / ``block_diag`` of several modules' parameters -- a fresh tensor every time, whose storage address the caching
    allocator recycles, so it must never be a cache key)"""

    def __init__(self, base, prefix: str, leaves):
        self.base, self.prefix, self.leaves = base, prefix, tuple(leaves)

    def get(self, key, params, build):
        return self.base.get(self.prefix + key, self.leaves, build)


class _StackedBN:
    """BatchNorm1d layers of equal width side by side (their columns concatenated); None = a padding block (gamma 1, beta 0, nothing
    tracked). Each layer keeps its own running buffers and batch counter."""

    def __init__(self, parts, width: int):
        self.parts, self.width = list(parts), width


def _bn_any(bn, mean, var, cnt):
    """_bn_train(..., want_rstd=True) for a BatchNorm1d or a _StackedBN -> (s, t, rstd)"""
    if not isinstance(bn, _StackedBN):
        return _bn_train(bn, mean, var, cnt, want_rstd=True)
    w, outs = bn.width, []
    for i, part in enumerate(bn.parts):
        m, v = mean[i * w:(i + 1) * w], var[i * w:(i + 1) * w]                 # (1-D slices: contiguous)
        if part is None:
            r = torch.rsqrt(v + 1e-5)
            outs.append((r, -m * r, r))
        else:
            outs.append(_bn_train(part, m, v, cnt, want_rstd=True))
    return tuple(torch.cat([o[k] for o in outs]).contiguous() for k in range(3))


class DenseTrain(torch.autograd.Function):
    """one ``Seq(Linear, ReLU, BatchNorm1d)`` of MLP() with batch statistics (models/basic_modules.py:31-36)"""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, bn, packs=_NoPacks):
        ops = get_ops()
        dev = x.device
        xa = _rows16(x)
        K, N = x.shape[1], weight.shape[0]
        pk = packs.get("fwd", (weight, bias), lambda: _pack_fwd(weight, bias, dev))
        y = _buf(xa.shape[0], N, dev)
        ops.gemm(Mat.of(xa, 0, K), pk, relu=True, Y=Mat.of(y, 0, N))
        mean, var, cnt, share = batch_moments(ops, Mat.of(y, 0, N))
        s, t, rstd = _bn_train(bn, mean, var, cnt, want_rstd=True)
        z = _buf(xa.shape[0], N, dev)
        ops.col_affine(Mat.of(y, 0, N), s, t, out=Mat.of(z, 0, N))
        ctx.save_for_backward(xa, weight, y, mean, rstd, gamma)
        ctx.dims = (K, N)
        ctx.share = share
        ctx.packs = packs
        return z[:, :N]

    @staticmethod
    def backward(ctx, dz):
        ops = get_ops()
        xa, weight, y, mean, rstd, gamma = ctx.saved_tensors
        K, N = ctx.dims
        dev = dz.device
        dza = _rows16(dz)
        Y = Mat.of(y, 0, N)
        sdz, sdzx = ops.bn_backward_stats(Mat.of(dza, 0, N), Y, mean, rstd)
        ksdz, ksdzx, _, _ = sync_backward_sums(sdz, sdzx, ctx.share)          # dgamma / dbeta stay per-rank partial sums (as dW)
        du = torch.empty_like(y) if y.shape[1] == N else torch.zeros_like(y)
        DU = Mat.of(du, 0, N)
        # (db = the column sums of du, from the pass that writes du)
        db = ops.bn_relu_backward(Mat.of(dza, 0, N), Y, mean, rstd, gamma.detach().float().contiguous(), ksdz, ksdzx, DU,
                                  want_sum=bool(ctx.needs_input_grad[2]))
        # a frozen layer (requires_grad False: e.g. train_deform_pose.py's frozen corr_extractor) pays for no weight-gradient GEMM
        dW = ops.gemm_tn(DU, Mat.of(xa, 0, K)) if ctx.needs_input_grad[1] else None
        dX = None
        if ctx.needs_input_grad[0]:
            wt = ctx.packs.get("wT", (weight,), lambda: _pack_bwd(weight.detach().t().contiguous(), dev))
            dX = _gemm_f32(ops, DU, wt, K)[:, :K]
        return dX, dW, db, (sdzx if ctx.needs_input_grad[3] else None), (sdz if ctx.needs_input_grad[4] else None), None, None


class NativeLinear(torch.autograd.Function):
    """a bare ``Linear`` (the last layer of mlp_transform, models/rignet.py:57): forward and both gradient GEMMs native"""

    @staticmethod
    def forward(ctx, x, weight, bias, packs=_NoPacks):
        ops = get_ops()
        dev = x.device
        xa = _rows16(x)
        K, N = x.shape[1], weight.shape[0]
        pk = packs.get("fwd", (weight, bias), lambda: _pack_fwd(weight, bias, dev))
        y = _buf(xa.shape[0], N, dev)
        ops.gemm(Mat.of(xa, 0, K), pk, relu=False, Y=Mat.of(y, 0, N))
        ctx.save_for_backward(xa, weight)
        ctx.dims = (K, N, bias is not None)
        ctx.packs = packs
        return y[:, :N]

    @staticmethod
    def backward(ctx, dy):
        ops = get_ops()
        xa, weight = ctx.saved_tensors
        K, N, has_bias = ctx.dims
        dya = _rows16(dy)
        DY = Mat.of(dya, 0, N)
        db = ops.bn_backward_stats(DY)[0] if (has_bias and ctx.needs_input_grad[2]) else None
        dW = ops.gemm_tn(DY, Mat.of(xa, 0, K)) if ctx.needs_input_grad[1] else None
        dX = None
        if ctx.needs_input_grad[0]:
            wt = ctx.packs.get("wT", (weight,), lambda: _pack_bwd(weight.detach().t().contiguous(), dy.device))
            dX = _gemm_f32(ops, DY, wt, K)[:, :K]
        return dX, dW, db, None


class EdgeMLPTrain(torch.autograd.Function):
    """per-edge ``MLP([2C, H, H])`` on [x_i ‖ x_j - x_i] with BatchNorm statistics over the edges and max over the incoming
    edges (models/basic_modules.py:153-155 / 192-195 in training mode); ``csr``: the unpadded CSR of the loop-normalised graph."""

    @staticmethod
    def forward(ctx, x, W1, b1, g1, be1, W2, b2, g2, be2, bn1, bn2, csr: CSR, packs=_NoPacks):
        assert not csr.quad, "batch statistics over edges need every edge exactly once"
        ops = get_ops()
        dev = x.device
        xa = _rows16(x)
        n, C = x.shape
        H = W1.shape[0]
        def vertex_pack():
            W1f = W1.detach().float()
            Wv = torch.cat([W1f[:, :C] - W1f[:, C:], W1f[:, C:]], 0)                  # [A | B] = x Wv^T + [b1 | 0]
            return _pack_fwd(Wv, torch.cat([b1.detach().float(), torch.zeros(H, device=W1.device)], 0), dev)
        vertex = packs.get("vertex", (W1, b1), vertex_pack)
        ab = _buf(n, 2 * H, dev)
        ops.gemm(Mat.of(xa, 0, C), vertex, relu=False, Y=Mat.of(ab, 0, 2 * H))
        A, B = Mat.of(ab, 0, H), Mat.of(ab, H, H)
        e_live = csr.rowptr[csr.n_nodes:csr.n_nodes + 1]
        z1 = _buf(csr.capacity, H, dev)
        loc1 = ops.edge_gather_relu(A, B, csr, Mat.of(z1, 0, H), want_stats=True)       # (its batch statistics from the same pass)
        mean1, var1, cnt, share1 = batch_moments(ops, Mat.of(z1, 0, H), rows_dev=e_live, local=loc1)
        s1, t1, rstd1 = _bn_any(bn1, mean1, var1, cnt)
        Hp, Kp = max(H, 32), (H + 31) // 32 * 32

        def w2_pack():
            W2p = torch.zeros((Hp, Kp), dtype=torch.float32, device=dev)
            W2p[:H, :H] = W2.detach().float()
            return W2p.contiguous(), _pad_to(b2.detach().float(), Hp, 0.0), torch.ones(Hp, device=dev), torch.zeros(Hp, device=dev)
        W2p, b2p, ones, zeros = packs.get("w2", (W2, b2), w2_pack)
        pe = packing.PackedEdge(H, _pad_to(s1, Kp, 1.0), _pad_to(t1, Kp, 0.0), W2p, b2p, ones, zeros, None)
        z2 = _buf(csr.capacity, H, dev)
        ops.edge_hidden(A, B, csr, pe, Mat.of(z2, 0, H))
        mean2, var2, cnt2, share2 = batch_moments(ops, Mat.of(z2, 0, H), rows_dev=e_live)
        s2, t2, rstd2 = _bn_any(bn2, mean2, var2, cnt2)
        out = _buf(n, H, dev)
        arg, zwin = ops.segmax_affine_arg(Mat.of(z2, 0, H), csr.rowptr, n, Mat.of(out, 0, H), s2, t2, want_zwin=True)
        ctx.save_for_backward(xa, W1, W2, z1, z2, mean1, rstd1, mean2, rstd2, g1, g2, s1, t1, arg, zwin, ab)
        ctx.csr = csr
        ctx.dims = (n, C, H)
        ctx.shares = (share1, share2)
        ctx.packs = packs
        return out[:, :H]

    @staticmethod
    def backward(ctx, dout):
        ops = get_ops()
        xa, W1, W2, z1, z2, mean1, rstd1, mean2, rstd2, g1, g2, s1, t1, arg, zwin, ab = ctx.saved_tensors
        csr: CSR = ctx.csr
        n, C, H = ctx.dims
        dev = dout.device
        e_live = csr.rowptr[csr.n_nodes:csr.n_nodes + 1]
        Z1, Z2 = Mat.of(z1, 0, H), Mat.of(z2, 0, H)
        do = _rows16(dout)
        DO = Mat.of(do, 0, H)
        # BatchNorm2 + ReLU behind the max: one-hot gradient per (vertex, channel)
        sdz2, sdzx2 = ops.segmax_bn_backward_stats(DO, arg, None, mean2, rstd2, zwin=zwin)   # (the winners' values were kept: no gather)
        k2, kx2, _, _ = sync_backward_sums(sdz2, sdzx2, ctx.shares[1])
        du2 = _buf(z2.shape[0], H, dev)                 # (the kernel zeroes the rows past E': they feed the dX GEMM below)
        DU2 = Mat.of(du2, 0, H)
        need = ctx.needs_input_grad                      # (x, W1, b1, g1, be1, W2, b2, g2, be2, ...): frozen layers skip their dW GEMMs
        db2 = ops.segmax_bn_relu_backward(DO, arg, Z2, csr.rowptr, csr.dst, mean2, rstd2, g2.detach().float().contiguous(), k2, kx2, DU2,
                                          want_sum=True)                               # db2 = column sums of du2, from the same pass
        # Linear2 on h = s1 Z1 + t1:  dW2 = du2^T h = (du2^T Z1) diag(s1) + db2 (x) t1. The product is taken on rows CENTRED on the layer's
        # batch mean, Mc = du2^T (Z1 - 1 mean1^T) = M - db2 (x) mean1: Z1 is post-ReLU (mean > 0, often >> std), and the BatchNorm sums
        # below need exactly M - db2 (x) mean1 -- formed after a bf16 x 3 contraction of the uncentred rows it cancelled heavily
        m1c = mean1.detach().float().contiguous()
        Mc = ops.gemm_tn(DU2, Z1, rows_dev=e_live, b_shift=m1c) if need[5] else None
        dW2 = ((Mc + db2[:, None] * m1c[None, :H]) * s1[None, :H] + db2[:, None] * t1[None, :H]) if need[5] else None
        w2t = ctx.packs.get("w2T", (W2,), lambda: _pack_bwd(W2.detach().t().contiguous(), dev))
        dh = _gemm_f32(ops, DU2, w2t, H)                                               # d(s1 Z1 + t1)  [capacity, H]
        DH = Mat.of(dh, 0, H)
        # BatchNorm1 + ReLU over the edges. Its two sums over dh = du2 W2 follow from M = du2^T Z1, db2 and W2 (H x H algebra in fp64):
        # no pass over the edge rows -- unless Linear2 is frozen and M was never formed
        if Mc is not None and not edge_sums_by_pass():
            sdz1, sdzx1 = ops.edge_bn_sums_from_products(Mc, db2, W2.detach().float().contiguous(), torch.zeros_like(m1c), rstd1)
        else:
            sdz1, sdzx1 = ops.bn_backward_stats(DH, Z1, mean1, rstd1, rows_dev=e_live)
        k1, kx1, _, _ = sync_backward_sums(sdz1, sdzx1, ctx.shares[0])
        # Z1 = relu(A[dst] + B[src]): the BatchNorm1 + ReLU gradient of an edge is evaluated where it is summed into dA[dst] / dB[src]
        # (never stored; both sums in a fixed order: the source side walks the transposed graph)
        dab = _buf(n, 2 * H, dev)
        # (Z1 is not read again: relu(A[dst] + B[src]) is rebuilt from the n x 2H matrix [A | B] the forward kept, out of the caches)
        ops.edge_bn_scatter_backward(DH, None, csr, n, Mat.of(dab, 0, H), Mat.of(dab, H, H), mean1, rstd1, g1.detach().float().contiguous(),
                                     k1, kx1, ZA=Mat.of(ab, 0, H), ZB=Mat.of(ab, H, H))
        DAB = Mat.of(dab, 0, 2 * H)
        db1 = ops.bn_backward_stats(Mat.of(dab, 0, H))[0] if need[2] else None
        dW1 = None
        if need[1]:
            dWv = ops.gemm_tn(DAB, Mat.of(xa, 0, C))                                   # [2H, C]
            dW1 = torch.cat([dWv[:H], dWv[H:] - dWv[:H]], 1)                           # back to [W_a | W_b]: W_v = [[W_a - W_b], [W_b]]
        dX = None
        if ctx.needs_input_grad[0]:                       # (positions and input features are leaves without a gradient)
            def wvt_pack():
                W1f = W1.detach().float()
                Wv = torch.cat([W1f[:, :C] - W1f[:, C:], W1f[:, C:]], 0)
                return _pack_bwd(Wv.t().contiguous(), dev)
            dX = _gemm_f32(ops, DAB, ctx.packs.get("wvT", (W1,), wvt_pack), C)[:, :C]
        return (dX, dW1, db1, (sdzx1 if need[3] else None), (sdz1 if need[4] else None), dW2, (db2 if need[6] else None),
                (sdzx2 if need[7] else None), (sdz2 if need[8] else None), None, None, None, None)


class SegMaxPool(torch.autograd.Function):
    """scatter_max over the vertices of each mesh (models/rignet.py:63): native arg-max forward, index routing backward"""

    @staticmethod
    def forward(ctx, x, mesh_ptr, n_graphs: int):
        ops = get_ops()
        xa = _rows16(x)
        C = x.shape[1]
        out = torch.zeros((n_graphs, _ld4(C)), dtype=torch.float32, device=x.device)
        arg = ops.segmax_affine_arg(Mat.of(xa, 0, C), mesh_ptr, n_graphs, Mat.of(out, 0, C))
        ctx.save_for_backward(arg)
        ctx.shape = (x.shape[0], C)
        return out[:, :C]

    @staticmethod
    def backward(ctx, dout):
        (arg,) = ctx.saved_tensors
        n, C = ctx.shape
        # one (row, column) target per (segment, column); empty segments (arg = -1) add 0 to row 0. No boolean indexing: that reads
        # the number of live entries back to the host (a stream synchronisation per pooling layer, ~11 ms of host stall each, measured)
        dx = torch.zeros((n, C), dtype=torch.float32, device=dout.device)
        live = arg >= 0
        dx.scatter_add_(0, arg.clamp(min=0).long(), torch.where(live, dout.float(), torch.zeros((), device=dout.device)))
        return dx, None, None


class RowGather(torch.autograd.Function):
    """``g[batch]`` (the per-mesh row repeated over the mesh's vertices, models/rignet.py:64). The backward is the segment sum
    ``onehot(batch)^T dout`` on morig_gemm_tn: products with 0 / 1 are exact and the row chunks are added in a fixed order, where
    torch's index backward sorts and serialises (5.4 ms per call on a 32 768 x 1024 gradient, measured)."""

    @staticmethod
    def forward(ctx, g, batch, n_graphs: int):
        ctx.save_for_backward(batch)
        ctx.ng = n_graphs
        return g.index_select(0, batch)

    @staticmethod
    def backward(ctx, dout):
        (batch,) = ctx.saved_tensors
        if ctx.ng > 256:                                   # many small meshes: the one-hot operand would dwarf the gradient itself
            return torch.zeros((ctx.ng, dout.shape[1]), dtype=torch.float32, device=dout.device).index_add_(0, batch, dout.float()), None, None
        ops = get_ops()
        onehot = torch.zeros((batch.shape[0], _ld4(ctx.ng)), dtype=torch.float32, device=dout.device)
        onehot.scatter_(1, batch.view(-1, 1), 1.0)
        do = _rows16(dout)
        return ops.gemm_tn(Mat.of(onehot, 0, ctx.ng), Mat.of(do, 0, dout.shape[1])), None, None


# ---- the reference's modules, composed from the blocks above (parameters are the module's own nn.Parameters) -------------------
def mlp_layer(x, layer):
    return DenseTrain.apply(x, layer[0].weight, layer[0].bias, layer[2].weight, layer[2].bias, layer[2], packs_of(layer))


def linear(x, lin):
    """a bare nn.Linear on the native forward / backward GEMMs"""
    return NativeLinear.apply(x, lin.weight, lin.bias, packs_of(lin))


def edge_mlp(x, csr: CSR, mlp):
    l1, l2 = mlp[0], mlp[1]
    return EdgeMLPTrain.apply(x, l1[0].weight, l1[0].bias, l1[2].weight, l1[2].bias, l2[0].weight, l2[0].bias, l2[2].weight, l2[2].bias,
                              l1[2], l2[2], csr, packs_of(mlp))


def edgeconvmotion(ec, pos, x, csr: CSR, pos_branch=None):
    """EdgeConvMotion (models/basic_modules.py:179-202): [nn_x branch | nn_pos branch], each max-aggregated; ``pos_branch``: the
    nn_pos result when it was computed with its siblings (``stacked_pos_branches``)"""
    return torch.cat([edge_mlp(x, csr, ec.nn_x), pos_branch if pos_branch is not None else edge_mlp(pos, csr, ec.nn_pos)], 1)


def stacked_pos_branches(net, pos, csr: CSR, which: str):
    """The nn_pos branches of the three EdgeConvMotions of a GCNRig on one graph take the same input (pos), the same graph and have
    the same shape (6 -> 16 -> 16): ONE edge MLP evaluates them together -- Linear1 rows stacked, Linear2 block-diagonal, BatchNorm
    columns side by side (statistics, running buffers and gradients stay per layer: a BatchNorm column never sees its neighbours),
    padded with an all-zero block to the 64 columns the edge kernels come in. As three 16-wide blocks they were 36 forward and 36
    backward blocks of ~30 launches per step, 8 ms of a 92 ms step for 1 % of its arithmetic. -> [three [n, 16] results], or None
    when the layers do not have that common shape (the caller then evaluates them one by one)."""
    mlps = [getattr(g, which).nn_pos for g in (net.gcu_1, net.gcu_2, net.gcu_3)]
    try:
        lins = [(m[0][0], m[0][2], m[1][0], m[1][2]) for m in mlps]
    except (IndexError, TypeError):
        return None
    H, C2 = lins[0][0].weight.shape
    W = 64
    ok = H * len(lins) <= W and W % H == 0 and all(
        l1.weight.shape == (H, C2) and l2.weight.shape == (H, H) and l1.bias is not None and l2.bias is not None and
        b1.weight is not None and b1.bias is not None and b2.weight is not None and b2.bias is not None and
        b1.eps == b2.eps == lins[0][1].eps for l1, b1, l2, b2 in lins)
    if not ok or C2 != 2 * pos.shape[1]:
        return None
    dev = pos.device
    pad = W - H * len(lins)
    cache = packs_of(net)
    z_w1, z_w2, z_v, o_v = cache.get(which + ":pos:pad", (), lambda: (
        torch.zeros((pad, C2), device=dev), torch.zeros((pad, pad), device=dev), torch.zeros(pad, device=dev), torch.ones(pad, device=dev)))
    if z_v.device != dev:
        return None
    cat = lambda ts, fill: torch.cat(list(ts) + ([fill] if pad else []))
    W1 = cat((l[0].weight for l in lins), z_w1)
    b1 = cat((l[0].bias for l in lins), z_v)
    g1, be1 = cat((l[1].weight for l in lins), o_v), cat((l[1].bias for l in lins), z_v)
    W2 = torch.block_diag(*([l[2].weight for l in lins] + ([z_w2] if pad else [])))
    b2 = cat((l[2].bias for l in lins), z_v)
    g2, be2 = cat((l[3].weight for l in lins), o_v), cat((l[3].bias for l in lins), z_v)
    nopad = [None] * (pad // H)
    bn1, bn2 = _StackedBN([l[1] for l in lins] + nopad, H), _StackedBN([l[3] for l in lins] + nopad, H)
    leaves = [q for l in lins for q in (l[0].weight, l[0].bias, l[2].weight, l[2].bias)]
    packs = cache if cache is _NoPacks else _KeyedPacks(cache, which + ":pos:", leaves)
    out = EdgeMLPTrain.apply(pos, W1, b1, g1, be1, W2, b2, g2, be2, bn1, bn2, csr, packs)
    return [out[:, i * H:(i + 1) * H] for i in range(len(lins))]


def gcumotion(gcu, pos, x, csr_tpl: CSR, csr_geo: CSR, pos_tpl=None, pos_geo=None):
    """GCUMotion (models/basic_modules.py:205-219)"""
    both = torch.cat([edgeconvmotion(gcu.edge_conv_tpl, pos, x, csr_tpl, pos_tpl), edgeconvmotion(gcu.edge_conv_geo, pos, x, csr_geo, pos_geo)], 1)
    return mlp_layer(both, gcu.mlp[0])


def gcnrig(net, pos, feature, csr_tpl: CSR, csr_geo: CSR, batch, mesh_ptr, n_graphs: int):
    """GCNRig.forward (models/rignet.py:59-67)"""
    tr = getattr(net, net.TRANSFORM)
    pt = stacked_pos_branches(net, pos, csr_tpl, "edge_conv_tpl") or [None] * 3
    pg = stacked_pos_branches(net, pos, csr_geo, "edge_conv_geo") or [None] * 3
    a = gcumotion(net.gcu_1, pos, feature, csr_tpl, csr_geo, pt[0], pg[0])
    b = gcumotion(net.gcu_2, pos, a, csr_tpl, csr_geo, pt[1], pg[1])
    c = gcumotion(net.gcu_3, pos, b, csr_tpl, csr_geo, pt[2], pg[2])
    g = SegMaxPool.apply(mlp_layer(torch.cat([a, b, c], 1), net.mlp_glb[0]), mesh_ptr, n_graphs)
    x5 = torch.cat([RowGather.apply(g, batch, n_graphs), pos, feature, a, b, c], 1)                              # repeat_interleave over sorted batch ids (:64)
    h = mlp_layer(mlp_layer(x5, tr[0][0]), tr[0][1])
    return linear(h, tr[1])


def temporal_attn(attn, x):
    """TemporalAttn.forward (models/rignet.py:36-46): a CLS token over the T keyframes, multi-head scaled dot-product attention,
    only token 0 kept -- 6 tokens per vertex, left to torch; the feed-forward MLP runs on the native blocks."""
    V = x.shape[0]
    nh = attn.num_heads
    tok = torch.cat([attn.cls_token.expand(V, -1, -1), x], 1)
    # The projections run on the native GEMMs (forward, dX and dW): as torch matmuls they were 6 ms of a 110 ms step on hipBLASLt's
    # float32 kernels. Only token 0 of the attention output is kept (:45) and token 0 is the CLS token of EVERY vertex, so the one
    # query row that matters is w_qs(cls_token), shared by all vertices, and only row 0 goes through w_o; keys and values need all
    # T + 1 tokens. The 1 x (T + 1) attention itself stays torch autograd.
    L = tok.shape[1]
    tok2 = tok.reshape(V * L, -1)
    kh = linear(tok2, attn.w_ks).reshape(V, L, nh, -1)
    vh = linear(tok2, attn.w_vs).reshape(V, L, nh, -1)
    q0 = linear(attn.cls_token.reshape(1, -1), attn.w_qs).reshape(nh, -1)
    # (broadcast products: as einsum / bmm these go to hipBLASLt float32 kernels that take 0.5-0.9 ms on such skinny shapes)
    att = torch.softmax((kh * q0).sum(-1) / math.sqrt(kh.shape[-1]), dim=1)                 # [V, L, nh], over the tokens
    res0 = (att.unsqueeze(-1) * vh).sum(1).reshape(V, -1)
    h = linear(res0, attn.w_o)
    return mlp_layer(mlp_layer(h, attn.feedforward[0]), attn.feedforward[1])


def canonical_csr(csr: CSR) -> CSR:
    """The CSR build claims a target's slots with an atomic cursor: the order of the edges INSIDE a segment changes from build to
    build. Max-aggregation does not see it; the training step does -- the segment sums of the backward add in row order, and the
    arg-max keeps the first of tied rows. Sorting every segment by source (one stable device sort of the live rows, stream-ordered,
    no host read) makes the order a function of the graph alone: with it a training step is bit-reproducible from run to run
    (MORIG_TRAIN_CANONICAL_CSR=0 skips the sort). Returns a NEW CSR (the argument is left as built)."""
    if os.environ.get("MORIG_TRAIN_CANONICAL_CSR", "1") == "0":
        return csr
    dev = csr.src.device
    n = csr.n_nodes
    live = csr.rowptr[n]
    pos = torch.arange(csr.capacity, device=dev)
    span = int(max(n, 1)) + 1
    big = torch.full((), span * span, dtype=torch.int64, device=dev)
    key = torch.where(pos < live, csr.dst.long().clamp(0, n) * span + csr.src.long().clamp(0, n), big)
    order = torch.sort(key, stable=True).indices
    return CSR(csr.rowptr, csr.src[order].contiguous(), csr.dst[order].contiguous(), csr.n_nodes, csr.capacity, csr.status,
               edge_count=csr.edge_count, quad=csr.quad)


# Canonical CSRs (and, through CSR.transposed(), their source-grouped twins) by the identity of the edge tensor they were built from:
# an epoch loop that feeds the SAME edge tensors again (a dataset cached on the device) then pays the four device sorts of a step
# once. Opt-in (MORIG_TRAIN_CSR_CACHE=1): bench.py's training step re-feeds one batch, and its time is meant to include the graph
# preparation a fresh batch costs. Entries die with their edge tensor (weak reference) or when it is written to (_version).
# REQUIREMENT while the switch is on: an edge tensor is IMMUTABLE once fed. A hit is validated by (data_ptr, shape, n, object identity,
# autograd version counter) only -- writes that bypass the counter (`edge_index.data[...] = ...`, a kernel writing through data_ptr(), an
# alias of the same storage edited in place) are NOT seen and the stale CSR and its transpose would be reused silently; a cached CSR's
# index-range status word is checked by the forward that built it, not again on a hit. After such an edit call ``invalidate_csr_cache()``.
_CSR_CACHE: dict = {}


def invalidate_csr_cache() -> None:
    """drop every cached canonical CSR (MORIG_TRAIN_CSR_CACHE=1): needed only after an edge tensor was edited in a way the autograd
    version counter does not see (see above)"""
    _CSR_CACHE.clear()


def _canonical_csr_of(edge_index: torch.Tensor, n: int) -> CSR:
    ops = get_ops()
    if os.environ.get("MORIG_TRAIN_CSR_CACHE", "0") != "1":
        return canonical_csr(ops.csr_build(edge_index, n))
    key = (edge_index.data_ptr(), tuple(edge_index.shape), n)
    hit = _CSR_CACHE.get(key)
    if hit is not None and hit[0]() is edge_index and hit[1] == edge_index._version:
        return hit[2]
    csr = canonical_csr(ops.csr_build(edge_index, n))
    for k in [k for k, v in _CSR_CACHE.items() if v[0]() is None]:
        del _CSR_CACHE[k]
    _CSR_CACHE[key] = (weakref.ref(edge_index), edge_index._version, csr)
    return csr


def graph_state(data):
    """CSRs of the two loop-normalised graphs + mesh offsets: built once per batch, shared by every block"""
    ops = get_ops()
    dev = data.pos.device
    n = data.pos.shape[0]
    ng = getattr(data, "num_graphs", None)
    if ng is None:
        ng = int(data.batch.max().item()) + 1
    counts = torch.bincount(data.batch, minlength=ng)
    mesh_ptr = torch.zeros(ng + 1, dtype=torch.int32, device=dev)
    mesh_ptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
    return dict(n=n, ng=int(ng), csr_tpl=_canonical_csr_of(data.tpl_edge_index, n),
                csr_geo=_canonical_csr_of(data.geo_edge_index, n), mesh_ptr=mesh_ptr, batch=data.batch.long())


def _motion_backbone(model, data, input_flow, st, aggr_method):
    """the keyframe loop + normalisation + aggregation shared by the three heads (models/rignet.py:82-98, 196-203): one
    motionNet pass PER keyframe, each with its own batch statistics, as the reference's Python loop does"""
    pos = data.pos.float()
    flow = input_flow.float()
    frames = []
    for t in range(model.num_keyframes):
        m = gcnrig(model.motionNet, pos, flow[:, 3 * t:3 * t + 3], st["csr_tpl"], st["csr_geo"], st["batch"], st["mesh_ptr"], st["ng"])
        frames.append(F.normalize(m, dim=1))
    motion_all = torch.stack(frames, 1)
    if aggr_method == "attn":
        aggr = temporal_attn(model.aggragator, motion_all)
    elif aggr_method == "mean":
        aggr = motion_all.mean(1)
    elif aggr_method == "max":
        aggr = motion_all.max(1)[0]
    else:
        raise NotImplementedError
    return motion_all, F.normalize(aggr, dim=1)


def _guarded_step(data, body, state=None, dev=None):
    """run ``body(st)`` on the model's device, ``st = state(data)`` (default: ``graph_state``). The CSR status words of the graph
    build are read BEFORE the body runs (an out-of-range edge index raises like the reference's index error, and nothing has
    touched the BatchNorm buffers yet); the split-fp16 range flag is cleared before and checked after (only
    MORIG_TRAIN_PRECISION=f16x3 can raise it -- by then the running buffers of this step have moved, as they would have in a step
    that ends in a NaN loss)."""
    ops = get_ops()
    dev = data.pos.device if dev is None else dev
    state = graph_state if state is None else state
    from .native import MorigNativeError
    with (torch.cuda.device(dev) if dev.type == "cuda" else contextlib.nullcontext()):
        flag = ops._flag(dev)
        flag.zero_()
        collect = getattr(ops, "_csr_status", None) is None and hasattr(ops, "_state")
        if collect:
            ops._csr_status = []
        try:
            st = state(data)
        finally:
            stats = None
            if collect:
                stats, ops._csr_status = ops._csr_status, None
        if stats and any(w != 0 for w in torch.cat(stats).tolist()):
            raise MorigNativeError("edge_index out of range for the vertex count it was built with (train-mode step)")
        out = body(st)
        if int(flag.item()) != 0:
            raise MorigNativeError("an operand left the split-fp16 range in the train-mode forward: unset MORIG_TRAIN_PRECISION "
                                   "(the default runs the train-mode contractions on the exact-fp32 MFMA kernels)")
    return out


def skinnet_inner(net, data, motion, st):
    """SkinNet_inner.forward (models/rignet.py:158-182)"""
    cols = torch.tensor(net.sample_columns(data.skin_input.shape[1]), dtype=torch.long, device=motion.device)
    raw = torch.cat([data.pos.float(), data.skin_input.float()[:, cols]], 1)
    x1 = gcumotion(net.gcu1, raw, motion, st["csr_tpl"], st["csr_geo"])
    g = SegMaxPool.apply(mlp_layer(mlp_layer(x1, net.multi_layer_tranform2[0]), net.multi_layer_tranform2[1]), st["mesh_ptr"], st["ng"])
    x2 = gcumotion(net.gcu2, raw, x1, st["csr_tpl"], st["csr_geo"])
    x3 = gcumotion(net.gcu3, raw, x2, st["csr_tpl"], st["csr_geo"])
    cb = net.cls_branch
    h = mlp_layer(mlp_layer(torch.cat([x3, RowGather.apply(g, st["batch"], st["ng"])], 1), cb[0][0]), cb[0][1])
    return linear(h, cb[1])


def skin_motion_step(model, data, input_flow):
    """SkinMotion.forward in training mode with an autograd graph (models/rignet.py:196-205) -> (motion_all, motion_aggr, skin_cls_pred)"""
    def body(st):
        motion_all, aggr = _motion_backbone(model, data, input_flow, st, "attn")
        return motion_all, aggr, skinnet_inner(model.skinNet, data, aggr, st)
    return _guarded_step(data, body)


def motion_head_step(model, data, input_flow):
    """JointNetMotion / MaskNetMotion.forward in training mode with an autograd graph (models/rignet.py:82-100):
    -> (motion_all [n, T, 32], motion_aggr, head output). Raises if an operand left the split-fp16 range on the way."""
    def body(st):
        motion_all, aggr = _motion_backbone(model, data, input_flow, st, model.aggr_method)
        head = getattr(model, model._head)
        out = gcnrig(head, data.pos.float(), aggr, st["csr_tpl"], st["csr_geo"], st["batch"], st["mesh_ptr"], st["ng"])
        return motion_all, aggr, out
    return _guarded_step(data, body)
