"""Host-side parameter packing for the HIP kernels (pure tensor algebra, device-agnostic).

What is restructured, exactly:
  * eval-mode BatchNorm -> per-channel affine  y = s*x + t,  s = gamma/sqrt(var+eps), t = beta - mean*s.
    MLP() is Linear -> ReLU -> BN (models/basic_modules.py:33), so the affine is applied AFTER the
    ReLU by the producing kernel's epilogue (and, inside the fused edge kernel, to the hidden layer
    while it is gathered). Nothing is folded across a ReLU.
  * first edge Linear per vertex:  W1 [x_i ; x_j - x_i] = (W1a - W1b) x_i + W1b x_j
    (message of EdgeConv / EdgeConvMotion, models/basic_modules.py:153-155, 192-195).
  * concatenations become column placement: weights are column-permuted once so activations can
    stay where their producer wrote them.
Weights are zero padded to the kernels' tile grid: rows to 32/64/128, K to a multiple of 32.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch
import torch.nn as nn

# Row normalisation [r06]. fp16 has a 5-bit exponent: the (hi, lo) split of a weight w carries 22 bits only while lo = fp16(w - hi) is a
# NORMAL half, i.e. |w| >= 0.125; below that lo is a subnormal with absolute precision 2^-25, so a row whose largest entry is m is carried
# to 2^-25 / m of its size: 21 bits for the rows of a He-initialised Linear (m ~ 0.08), 12 bits for a row folded with a small BatchNorm
# scale (W2 diag(s1), m ~ 1e-4), and a row above 65 000 has no image at all (the exact path for ever). A row is therefore multiplied by the
# power of two that puts its largest entry into [4096, 8192) -- exact -- and the factor is undone where it costs nothing:
# s act(W x + b) + t = (s / f) act((f W) x + f b) + t for f > 0, so bias and BN scale absorb it at pack time and no kernel changes.
# Where it is applied (MORIG_PACK_NORMALISE): "auto" (default) -- only a layer with a row that needs it (largest entry < 2^-6: fewer than
# 19 bits; or >= 2^14: about to leave the range); "all"; "0" = off. Why not everywhere, when it is exact and free of kernel changes:
# measured on the headline (tools/gpu_r06w*.sh, same box, interleaved) "all" costs 0.3-0.7 % of the step and lowers the network-level
# errors of the goldens by 0-45 % (4.2e-6 -> 3.2e-6 on the headline mesh) -- the bias-only [A | B] producers gain a scale vector to read,
# and the EdgeConv kernel itself runs 1 % slower on normalised W2 images: with every lo half a NORMAL fp16 number its mantissa bits are
# all live, the MFMA operands toggle more, and the kernel sits on the power cap. Ordinary layers already carry 20-21 bits.
PACK_NORMALISE = os.environ.get("MORIG_PACK_NORMALISE", "auto")
PACK_NORMALISE = {"1": "all", "": "auto"}.get(PACK_NORMALISE, PACK_NORMALISE)
assert PACK_NORMALISE in ("auto", "all", "0"), PACK_NORMALISE
_ROW_TARGET = 8192.0


def _rows_need_normalising(W: torch.Tensor) -> bool:
    m = W.detach().abs().amax(dim=1)
    m = m[m > 0]
    return bool(m.numel()) and (not bool(torch.isfinite(m).all()) or float(m.min()) < 2.0 ** -6 or float(m.max()) >= 2.0 ** 14)


def _row_factors(W: torch.Tensor) -> torch.Tensor:
    """per row the power of two f with max|row| * f in [4096, 8192) (1 for an all-zero or non-finite row)"""
    m = W.detach().abs().amax(dim=1).double()
    e = torch.floor(torch.log2(_ROW_TARGET / m.clamp_min(1e-300)))
    ok = torch.isfinite(m) & (m > 0)
    e = torch.where(ok, e, torch.zeros_like(e)).clamp(-120.0, 120.0)
    return torch.exp2(e).float()


def _roundup(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def col_tile(n: int) -> int:
    """column tile the GEMM launcher picks for a logical width n (morig_gemm in tile_gemm.hip)."""
    return 128 if n > 64 else (64 if n > 32 else 32)


def bn_affine(bn: nn.BatchNorm1d):
    s = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
    t = bn.bias.detach().float() - bn.running_mean.detach().float() * s
    return s, t


@dataclass
class PackedLinear:
    W: torch.Tensor                     # [Npad, Kpad]
    bias: Optional[torch.Tensor]        # [Npad]
    scale: Optional[torch.Tensor]       # [Npad]   (BN eval affine; None = identity)
    shift: Optional[torch.Tensor]
    N: int
    K: int
    Wsplit: Optional[torch.Tensor] = None   # split-fp16 image of W (same shape/stride), None = fp32 only
    Wsplit_bf16: Optional[torch.Tensor] = None   # split-bf16 image (gradient contractions, train_backward.py): takes precedence when set
    row_factor: Optional[torch.Tensor] = None    # [Npad] the power of two every row of W (and bias) was multiplied by; scale carries 1 / it


@dataclass
class PackedEdge:
    """second half of an edge MLP: hidden affine (BN1), Linear2, BN2; H = width."""
    H: int
    s1: Optional[torch.Tensor]          # None: the hidden affine is folded into W2 / b2
    t1: Optional[torch.Tensor]
    W2: torch.Tensor                    # [max(H,32), roundup(H,32)]
    b2: torch.Tensor
    s2: torch.Tensor
    t2: torch.Tensor
    W2split: Optional[torch.Tensor] = None


F16_SAFE_MAX = 6.0e4


def split_f16(Wp: torch.Tensor) -> Optional[torch.Tensor]:
    """Split-fp16 image of a packed weight matrix [Npad, Kpad] (Kpad % 32 == 0): every 32-float chunk of
    a row becomes [32 halves hi | 32 halves lo] with hi = fp16(w), lo = fp16(w - hi); same bytes, so the
    kernels' weight loader is unchanged (include/morig_hip.h, "split-fp16"). None if a weight does not
    fit the fp16 range."""
    N, Kp = Wp.shape
    assert Kp % 32 == 0
    if not bool(torch.isfinite(Wp).all()) or float(Wp.abs().max()) >= F16_SAFE_MAX:
        return None
    hi = Wp.half()
    lo = (Wp - hi.float()).half()
    img = torch.stack([hi.view(N, Kp // 32, 32), lo.view(N, Kp // 32, 32)], dim=2)     # [N, Kc, 2, 32] halves
    return img.reshape(N, Kp * 2).contiguous().view(torch.float32)


def split_bf16(Wp: torch.Tensor) -> torch.Tensor:
    """Split-bf16 image of a packed weight matrix, same layout as ``split_f16`` with bf16 halves: hi = bf16(w), lo = bf16(w - hi), both
    rounded to nearest even (what v_cvt_pk_bf16_f32 does to the activations in the kernel). bf16 has float32's exponent range: no
    range condition (the operands of the backward contractions are gradients)."""
    N, Kp = Wp.shape
    assert Kp % 32 == 0
    hi = Wp.to(torch.bfloat16)
    lo = (Wp - hi.float()).to(torch.bfloat16)
    img = torch.stack([hi.view(N, Kp // 32, 32), lo.view(N, Kp // 32, 32)], dim=2)     # [N, Kc, 2, 32] halves
    return img.reshape(N, Kp * 2).contiguous().view(torch.float32)


def _pad_vec(v: Optional[torch.Tensor], n: int, fill: float = 0.0) -> Optional[torch.Tensor]:
    if v is None:
        return None
    out = torch.full((n,), fill, dtype=torch.float32, device=v.device)
    out[: v.numel()] = v.float()
    return out.contiguous()


def unsplit_f16(buf: torch.Tensor, cols: int) -> torch.Tensor:
    """inverse of the split-fp16 activation layout (tests / debugging): fp32 values hi + lo of the first ``cols``
    logical columns of a [rows, ld] float32-typed buffer (ld % 32 == 0)."""
    rows, ld = buf.shape
    h = buf.contiguous().view(torch.float16).view(rows, ld // 32, 2, 32).float()
    return (h[:, :, 0, :] + h[:, :, 1, :]).reshape(rows, ld)[:, :cols]


def fold_hidden_affine(W2: torch.Tensor, b2: torch.Tensor, s1: torch.Tensor, t1: torch.Tensor):
    """W2 (s1*h + t1) + b2 = (W2 diag(s1)) h + (b2 + W2 t1): the BatchNorm that follows the FIRST edge ReLU is a
    linear map in front of Linear2, so it folds into Linear2 exactly (evaluated in fp64, rounded once)."""
    W = W2.double()
    return (W * s1.double()[None, :]).float(), (b2.double() + W @ t1.double()).float()


def pack_linear(weight: torch.Tensor, bias: Optional[torch.Tensor] = None, bn: Optional[nn.BatchNorm1d] = None,
                in_cols: Optional[Sequence[int]] = None, k_total: Optional[int] = None,
                row_tile: Optional[int] = None, split: bool = True) -> PackedLinear:
    """weight [N, K_src]. ``in_cols[j]`` = position of source column j in the kernel's input row
    (default identity); ``k_total`` = logical K of the kernel input (>= max(in_cols)+1). ``split=False``: no split-fp16 image
    (the exact-fp32 MFMA kernels only; saves the range check's host synchronisation -- the training path packs every step)."""
    W = weight.detach().float()
    N, Ksrc = W.shape
    identity = in_cols is None
    if in_cols is None:
        in_cols = list(range(Ksrc))
    K = max(in_cols) + 1 if k_total is None else k_total
    Npad = _roundup(N, row_tile or col_tile(N))
    Kpad = _roundup(K, 64)                      # 64: the split-fp16 dense kernel stages 64-deep K chunks
    Wp = torch.zeros((Npad, Kpad), dtype=torch.float32, device=W.device)
    if identity:
        Wp[:N, :Ksrc] = W                          # a plain copy (the training path packs every step)
    else:
        Wp[:N, torch.as_tensor(list(in_cols), device=W.device)] = W
    s = t = None
    if bn is not None:
        s, t = bn_affine(bn)
    Wp = Wp.contiguous()
    bp = _pad_vec(bias.detach() if bias is not None else torch.zeros(N, device=W.device), Npad)
    sp, tp, rf = _pad_vec(s, Npad, 1.0), _pad_vec(t, Npad), None
    if split and (PACK_NORMALISE == "all" or (PACK_NORMALISE == "auto" and _rows_need_normalising(Wp))):
        rf = _row_factors(Wp)
        Wp = (Wp * rf[:, None]).contiguous()
        bp = bp * rf
        sp = (sp if sp is not None else torch.ones(Npad, dtype=torch.float32, device=W.device)) / rf
        tp = tp if tp is not None else torch.zeros(Npad, dtype=torch.float32, device=W.device)
    return PackedLinear(Wp, bp, sp, tp, N, K, split_f16(Wp) if split else None, None, rf)


def couple_rowbias(g: PackedLinear, main: PackedLinear) -> PackedLinear:
    """`g` produces the per-segment row bias that `main` adds in front of its activation (morig_gemm: act(X W^T + bias + rowbias[seg])):
    the row bias has to arrive in main's normalised row units, so g's output scale takes main's row factors on top of undoing its own."""
    if main.row_factor is None:
        return g
    one = torch.ones_like(main.row_factor)
    gs = g.scale if g.scale is not None else one[: g.W.shape[0]]
    n = min(gs.numel(), main.row_factor.numel())
    gs = gs.clone()
    gs[:n] = gs[:n] * main.row_factor[:n]
    shift = g.shift if g.shift is not None else torch.zeros_like(gs)
    return PackedLinear(g.W, g.bias, gs, shift, g.N, g.K, g.Wsplit, g.Wsplit_bf16, g.row_factor)


def _normalise_edge(W2: torch.Tensor, b2: torch.Tensor, s2: torch.Tensor):
    """row normalisation of the second edge Linear: (f W2, f b2, s2 / f); the kernels' sign trick reads sign(s2), which f > 0 leaves"""
    if PACK_NORMALISE == "0" or (PACK_NORMALISE == "auto" and not _rows_need_normalising(W2)):
        return W2, b2, s2
    rf = _row_factors(W2)
    return (W2 * rf[:, None]).contiguous(), b2 * rf, s2 / rf


def pack_mlp_layer(layer: nn.Sequential, **kw) -> PackedLinear:
    """one ``Seq(Linear, ReLU, BN)`` of MLP()."""
    return pack_linear(layer[0].weight, layer[0].bias, layer[2], **kw)


def pack_edge_pair(mlps: Sequence[nn.Sequential]):
    """Edge MLPs ``MLP([2C, H, H])`` of several graphs sharing the same vertex input.
    Returns (vertex PackedLinear producing [A_0 | B_0 | A_1 | B_1 ...] of width 2*H*len, [PackedEdge])."""
    rows, biases, edges = [], [], []
    H = mlps[0][0][0].weight.shape[0]
    for m in mlps:
        lin1, bn1 = m[0][0], m[0][2]
        lin2, bn2 = m[1][0], m[1][2]
        W1 = lin1.weight.detach().float()
        C = W1.shape[1] // 2
        Wa, Wb = W1[:, :C], W1[:, C:]
        rows += [Wa - Wb, Wb]
        biases += [lin1.bias.detach().float(), torch.zeros_like(lin1.bias, dtype=torch.float32)]
        s1, t1 = bn_affine(bn1)
        s2, t2 = bn_affine(bn2)
        Hp, Kp = max(H, 32), _roundup(H, 32)
        Wf, bf = fold_hidden_affine(lin2.weight.detach().float(), lin2.bias.detach().float(), s1, t1)
        W2 = torch.zeros((Hp, Kp), dtype=torch.float32, device=W1.device)
        W2[:H, :H] = Wf
        W2, b2p, s2p = _normalise_edge(W2.contiguous(), _pad_vec(bf, Hp), _pad_vec(s2, Hp, 1.0)) if H >= 32 else \
            (W2.contiguous(), _pad_vec(bf, Hp), _pad_vec(s2, Hp, 1.0))
        edges.append(PackedEdge(H, None, None, W2, b2p, s2p, _pad_vec(t2, Hp), split_f16(W2) if H >= 32 else None))
    vertex = pack_linear(torch.cat(rows, 0), torch.cat(biases, 0))
    return vertex, edges


def pack_first_x3(A: torch.Tensor, B: torch.Tensor, b: torch.Tensor):
    """First edge Linear on a 3-channel input for morig_edgeconv_x3: (W1a [32, 4], W1b [32, 4], b1 [32]) from A = W_a - W_b, B = W_b
    ([H <= 32, 3]) and the bias; rows past H and column 3 are zero."""
    H = A.shape[0]
    assert A.shape == (H, 3) and B.shape == (H, 3) and H <= 32
    W1a = torch.zeros((32, 4), dtype=torch.float32, device=A.device)
    W1b = torch.zeros((32, 4), dtype=torch.float32, device=A.device)
    b1 = torch.zeros(32, dtype=torch.float32, device=A.device)
    W1a[:H, :3], W1b[:H, :3], b1[:H] = A, B, b
    return W1a.contiguous(), W1b.contiguous(), b1


def pack_pos_groups(units):
    """The POSITION branches (``nn_pos`` = MLP([2P, D, D]), D = 16) of several GCUMotion units that see the same positions and the
    same two graphs (models/basic_modules.py:193-195, 212-215: the motion and the head network of a rig model, rignet.py:82-99).
    Units are taken two at a time; a pair becomes ONE 32-wide edge layer per graph whose second Linear is block-diagonal:
      vertex PackedLinear: pos -> per pair [A_tpl(2D) | B_tpl(2D) | A_geo(2D) | B_geo(2D)], A / B as pack_edge_pair builds them;
      per pair (PackedEdge tpl, PackedEdge geo) of width 2D.
    Six 16-wide EdgeConvs per graph (half of a 32-column MFMA tile each, 64-byte gathers) become three 32-wide ones.
    Returns (vertex, [(edge_tpl, edge_geo)], n_pairs); an odd last unit is not covered (the caller runs it on its own)."""
    n_pairs = len(units) // 2
    if n_pairs == 0:
        return None, [], 0
    rows, biases, edges, firsts = [], [], [], []
    D = units[0].edge_conv_tpl.nn_pos[0][0].weight.shape[0]
    for g in range(n_pairs):
        pair = units[2 * g: 2 * g + 2]
        pe = []
        for which in ("edge_conv_tpl", "edge_conv_geo"):
            mlps = [getattr(u, which).nn_pos for u in pair]
            A, B, bA, W2s, b2s, s2s, t2s = [], [], [], [], [], [], []
            for m in mlps:
                lin1, bn1, lin2, bn2 = m[0][0], m[0][2], m[1][0], m[1][2]
                W1 = lin1.weight.detach().float()
                assert W1.shape[0] == D and lin2.weight.shape == (D, D)
                C = W1.shape[1] // 2
                A.append(W1[:, :C] - W1[:, C:]); B.append(W1[:, C:]); bA.append(lin1.bias.detach().float())
                s1, t1 = bn_affine(bn1)
                s2, t2 = bn_affine(bn2)
                Wf, bf = fold_hidden_affine(lin2.weight.detach().float(), lin2.bias.detach().float(), s1, t1)
                W2s.append(Wf); b2s.append(bf); s2s.append(s2); t2s.append(t2)
            rows += [torch.cat(A, 0), torch.cat(B, 0)]
            biases += [torch.cat(bA, 0), torch.zeros(2 * D, dtype=torch.float32, device=rows[-1].device)]
            firsts.append(pack_first_x3(torch.cat(A, 0), torch.cat(B, 0), torch.cat(bA, 0)) if A[0].shape[1] == 3 else None)
            H = 2 * D
            W2 = torch.zeros((max(H, 32), _roundup(H, 32)), dtype=torch.float32, device=rows[-1].device)
            W2[:D, :D] = W2s[0]
            W2[D:H, D:H] = W2s[1]
            Hp = max(H, 32)
            W2, b2p, s2p = W2.contiguous(), _pad_vec(torch.cat(b2s), Hp), _pad_vec(torch.cat(s2s), Hp, 1.0)
            if H >= 32:
                W2, b2p, s2p = _normalise_edge(W2, b2p, s2p)
            pe.append(PackedEdge(H, None, None, W2, b2p, s2p, _pad_vec(torch.cat(t2s), Hp), split_f16(W2) if H >= 32 else None))
        # (et, eg, first_tpl, first_geo): `first_*` = the first Linear in the form morig_edgeconv_x3 evaluates in its loader (P == 3)
        edges.append((pe[0], pe[1], firsts[-2], firsts[-1]))
    vertex = pack_linear(torch.cat(rows, 0), torch.cat(biases, 0))
    return vertex, edges, n_pairs


FUSED_POINTCONV_WIDTHS = ((32, 64), (64, 128))        # (H, H3) pairs morig_pointconv_fused is instantiated for


def pack_pointconv(local_nn: nn.Sequential, cx: int):
    """PointConv local_nn = MLP([cx+3, H, H, H3]) on [x_j ‖ pos_j - pos_i] (PyG PointConv.message):
    -> source linear over [x | pos] (K = cx+3, with bias), target linear -W1p over the centre position,
       PackedEdge for BN1/Linear2/BN2, PackedLinear for Linear3/BN3."""
    l1, l2, l3 = local_nn[0], local_nn[1], local_nn[2]
    W1 = l1[0].weight.detach().float()
    H = W1.shape[0]
    assert W1.shape[1] == cx + 3 and l2[0].weight.shape == (H, H), "PointConv hidden layers must be square"
    src = pack_linear(W1, l1[0].bias)
    tgt = pack_linear(-W1[:, cx:], None)
    s1, t1 = bn_affine(l1[2])
    s2, t2 = bn_affine(l2[2])
    Hp, Kp = max(H, 32), _roundup(H, 32)
    Wf, bf = fold_hidden_affine(l2[0].weight.detach().float(), l2[0].bias.detach().float(), s1, t1)
    W2 = torch.zeros((Hp, Kp), dtype=torch.float32, device=W1.device)
    W2[:H, :H] = Wf
    W2 = W2.contiguous()
    edge = PackedEdge(H, None, None, W2, _pad_vec(bf, Hp), _pad_vec(s2, Hp, 1.0), _pad_vec(t2, Hp),
                      split_f16(W2) if H >= 32 else None)
    # fused kernel (csrc/pointconv_fused.hip): BN2 folds into Linear3 exactly as BN1 folds into Linear2 above
    fused = None
    H3 = l3[0].weight.shape[0]
    if (H, H3) in FUSED_POINTCONV_WIDTHS:
        W3f, b3f = fold_hidden_affine(l3[0].weight.detach().float(), l3[0].bias.detach().float(), s2, t2)
        s3, t3 = bn_affine(l3[2])
        fused = PackedLinear(W3f.contiguous(), b3f.contiguous(), s3.contiguous(), t3.contiguous(), H3, H, split_f16(W3f.contiguous()))
    return dict(src=src, tgt=tgt, edge=edge, last=pack_mlp_layer(l3), fused=fused)


def scale_additive(obj, c: float):
    """A (nested) packed structure with every ADDITIVE constant multiplied by c (a power of two: exact) -- biases, BatchNorm shifts, the
    first-layer bias of a ``pack_first_x3`` triple -- and everything multiplicative left alone. The GCU / GCNRig stacks are positively
    homogeneous of degree one in (inputs, additive constants) jointly: Linear, ReLU, the BN scale, max aggregation and max pooling all
    commute with a positive factor, so a stack whose inputs AND additive constants are scaled by c produces c times its outputs. That is
    the range shift of ``NativeModule`` (activations beyond the split-fp16 range: run the stack at 2^-k, multiply the result by 2^k)."""
    if c == 1.0:
        return obj
    if isinstance(obj, PackedLinear):
        return PackedLinear(obj.W, None if obj.bias is None else obj.bias * c, obj.scale, None if obj.shift is None else obj.shift * c,
                            obj.N, obj.K, obj.Wsplit, obj.Wsplit_bf16, obj.row_factor)
    if isinstance(obj, PackedEdge):
        return PackedEdge(obj.H, obj.s1, None if obj.t1 is None else obj.t1 * c, obj.W2, obj.b2 * c, obj.s2, obj.t2 * c, obj.W2split)
    if isinstance(obj, dict):
        return {k: scale_additive(v, c) for k, v in obj.items()}
    if isinstance(obj, tuple) and len(obj) == 3 and all(isinstance(t, torch.Tensor) for t in obj) and \
            tuple(obj[0].shape) == (32, 4) and tuple(obj[1].shape) == (32, 4) and tuple(obj[2].shape) == (32,):
        return (obj[0], obj[1], obj[2] * c)                 # pack_first_x3: (W1a, W1b, b1)
    if isinstance(obj, (list, tuple)):
        return type(obj)(scale_additive(v, c) for v in obj)
    return obj


def to_device(obj, device):
    """move a (nested) packed structure to ``device``."""
    if isinstance(obj, torch.Tensor):
        return obj.to(device)
    if isinstance(obj, (PackedLinear, PackedEdge)):
        return type(obj)(**{k: to_device(v, device) for k, v in obj.__dict__.items()})
    if isinstance(obj, dict):
        return {k: to_device(v, device) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(to_device(v, device) for v in obj)
    return obj
