"""Drop-in mirror of the reference's ``models/basic_modules.py`` graph blocks: same class names,
constructor signatures, ``forward()`` signatures and ``state_dict`` keys
(/root/reference/models/basic_modules.py:31-36, 142-219); the eval-mode arithmetic runs in the HIP
kernels of libmorig_hip.so through morig_amd.native (no PyG / torch_scatter / torch_cluster).

The torch containers built by ``MLP`` only HOLD parameters (so checkpoints load unchanged); they
are never called on the native path.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch.nn import BatchNorm1d as BN, Linear as Lin, ReLU, Sequential as Seq

from .. import packing
from ..native import Mat
from ..runtime import get_ops

__all__ = ["MLP", "EdgeConv", "GCU", "EdgeConvMotion", "GCUMotion", "NativeModule"]


def MLP(channels, batch_norm=True):
    """Parameter container with the reference's key layout ``{i}.0`` Linear / ``{i}.2`` BatchNorm1d
    (Linear -> ReLU -> BN per layer; models/basic_modules.py:31-36)."""
    if not batch_norm:
        return Seq(*[Seq(Lin(a, b), ReLU()) for a, b in zip(channels[:-1], channels[1:])])
    return Seq(*[Seq(Lin(a, b), ReLU(), BN(b, momentum=0.1)) for a, b in zip(channels[:-1], channels[1:])])


def _invalidate_packed(module, incompatible_keys):
    module._drop_packed()


class NativeModule(torch.nn.Module):
    """Base class: caches packed (kernel-layout) parameters and invalidates the cache whenever the
    parameters can have changed (``load_state_dict``, ``.to()/.cuda()``, ``train()``)."""

    def __init__(self):
        super().__init__()
        self._packed = None
        self._packed_device = None
        self.register_load_state_dict_post_hook(_invalidate_packed)

    def _drop_packed(self):
        for m in self.modules():
            if isinstance(m, NativeModule):
                m._packed = None

    def _apply(self, fn, *a, **kw):
        self._drop_packed()
        return super()._apply(fn, *a, **kw)

    def train(self, mode: bool = True):
        self._drop_packed()
        return super().train(mode)

    def packed(self, device):
        if self._packed is None or self._packed_device != device:
            self._packed = packing.to_device(self._pack(), device)
            self._packed_device = device
        return self._packed

    def _pack(self):
        raise NotImplementedError

    def forward(self, *args, **kwargs):
        """eval-mode forward on the HIP path. The whole forward runs inside the op layer's precision
        guard: split-fp16 MFMA first; if a kernel saw an operand outside the fp16 range the forward is
        repeated with fp32 MFMA (morig_amd.native.NativeOps.guarded)."""
        self._require_eval()
        ops = get_ops()
        dev = next(self.parameters()).device
        rng = torch.get_rng_state()                    # random FPS starts: a repeated attempt draws the same ones

        def attempt():
            torch.set_rng_state(rng)
            return self._forward(*args, **kwargs)
        return ops.guarded(dev, attempt)

    def _forward(self, *args, **kwargs):
        raise NotImplementedError

    def _require_eval(self):
        if self.training:
            raise NotImplementedError(
                f"{type(self).__name__}: only the eval-mode forward runs on the MI355X-native path "
                "(train-mode BatchNorm statistics over edges are out of scope, SURVEY.md 8(f-4)); call model.eval()")


def _as_matrix(x: torch.Tensor) -> torch.Tensor:
    x = x.unsqueeze(-1) if x.dim() == 1 else x
    return x.float().contiguous()


def _padded_copy(ops, x: torch.Tensor) -> torch.Tensor:
    """copy an arbitrary-width fp32 matrix into a buffer whose row stride is a multiple of 4 floats
    (GEMM operand alignment); padding columns are zero."""
    n, c = x.shape
    ld = (c + 3) // 4 * 4
    if ld == c and x.is_contiguous():
        return x
    buf = torch.zeros((n, ld), dtype=torch.float32, device=x.device)
    ops.copy2d(Mat.of(x), Mat.of(buf, 0, c))
    return buf


class EdgeConv(NativeModule):
    """models/basic_modules.py:142-163 -- message nn_pos([x_i ‖ x_j - x_i]), max over incoming edges."""

    def __init__(self, nn_pos, aggr="max", **kwargs):
        super().__init__()
        assert aggr == "max", "only aggr='max' is used by MoRig"
        self.nn_pos = nn_pos

    def _pack(self):
        vertex, (edge,) = packing.pack_edge_pair([self.nn_pos])
        return dict(vertex=vertex, edge=edge)

    def _forward(self, x, edge_index):
        ops = get_ops()
        x = _padded_copy(ops, _as_matrix(x))
        pk = self.packed(x.device)
        n, H = x.shape[0], pk["edge"].H
        csr = ops.csr_build(edge_index, n)
        ab = ops.empty(n, 2 * H, x.device)
        ops.gemm(Mat.of(x, 0, pk["vertex"].K), pk["vertex"], relu=False, Y=Mat.of(ab))
        out = ops.empty(n, H, x.device)
        ops.edgeconv(Mat.of(ab, 0, H), Mat.of(ab, H, H), csr, pk["edge"], Mat.of(out))
        return out


class EdgeConvMotion(NativeModule):
    """models/basic_modules.py:179-202 -- message nn_x([x_i ‖ x_j-x_i]) ‖ nn_pos([pos_i ‖ pos_j-pos_i])."""

    def __init__(self, nn_x, nn_pos, aggr="max", **kwargs):
        super().__init__()
        assert aggr == "max", "only aggr='max' is used by MoRig"
        self.nn_x = nn_x
        self.nn_pos = nn_pos

    def _pack(self):
        vx, (ex,) = packing.pack_edge_pair([self.nn_x])
        vp, (ep,) = packing.pack_edge_pair([self.nn_pos])
        return dict(vx=vx, ex=ex, vp=vp, ep=ep)

    def _forward(self, pos, x, edge_index):
        ops = get_ops()
        x = _padded_copy(ops, _as_matrix(x))
        pos = _padded_copy(ops, _as_matrix(pos))
        pk = self.packed(x.device)
        n, H, D = x.shape[0], pk["ex"].H, pk["ep"].H
        csr = ops.csr_build(edge_index, n)
        abx = ops.empty(n, 2 * H, x.device)
        abp = ops.empty(n, 2 * D, x.device)
        ops.gemm(Mat.of(x, 0, pk["vx"].K), pk["vx"], relu=False, Y=Mat.of(abx))
        ops.gemm(Mat.of(pos, 0, pk["vp"].K), pk["vp"], relu=False, Y=Mat.of(abp))
        out = ops.empty(n, H + D, x.device)
        ops.edgeconv(Mat.of(abx, 0, H), Mat.of(abx, H, H), csr, pk["ex"], Mat.of(out, 0, H))
        ops.edgeconv(Mat.of(abp, 0, D), Mat.of(abp, D, D), csr, pk["ep"], Mat.of(out, H, D))
        return out


class GCU(NativeModule):
    """models/basic_modules.py:165-177."""

    def __init__(self, in_channels, out_channels, aggr="max"):
        super().__init__()
        self.edge_conv_tpl = EdgeConv(nn_pos=MLP([in_channels * 2, out_channels // 2, out_channels // 2]), aggr=aggr)
        self.edge_conv_geo = EdgeConv(nn_pos=MLP([in_channels * 2, out_channels // 2, out_channels // 2]), aggr=aggr)
        self.mlp = MLP([out_channels, out_channels])

    def _pack(self):
        vertex, (et, eg) = packing.pack_edge_pair([self.edge_conv_tpl.nn_pos, self.edge_conv_geo.nn_pos])
        return dict(vertex=vertex, et=et, eg=eg, mlp=packing.pack_mlp_layer(self.mlp[0]))

    def run(self, ops, x: Mat, csr_tpl, csr_geo, out: Mat):
        """x: [n, C] window (16-byte aligned rows) -> out: [n, O] window."""
        pk = self.packed(x.base.device)
        n, H = x.rows, pk["et"].H
        dev = x.base.device
        ab = ops.empty(n, 4 * H, dev)
        ops.gemm(x, pk["vertex"], relu=False, Y=Mat.of(ab))
        ec = ops.empty(n, 2 * H, dev)
        ops.edgeconv(Mat.of(ab, 0, H), Mat.of(ab, H, H), csr_tpl, pk["et"], Mat.of(ec, 0, H))
        ops.edgeconv(Mat.of(ab, 2 * H, H), Mat.of(ab, 3 * H, H), csr_geo, pk["eg"], Mat.of(ec, H, H))
        ops.gemm(Mat.of(ec), pk["mlp"], relu=True, Y=out)

    def _forward(self, pos, tpl_edge_index, geo_edge_index):
        ops = get_ops()
        x = _padded_copy(ops, _as_matrix(pos))
        n = x.shape[0]
        pk = self.packed(x.device)
        out = ops.empty(n, pk["mlp"].N, x.device)
        self.run(ops, Mat.of(x, 0, pk["vertex"].K), ops.csr_build(tpl_edge_index, n), ops.csr_build(geo_edge_index, n), Mat.of(out))
        return out


class GCUMotion(NativeModule):
    """models/basic_modules.py:205-219. ``run`` supports R keyframe replicas of the feature with ONE
    position branch (nn_pos depends only on pos and the graph: models/rignet.py:85-86)."""

    def __init__(self, in_channels, out_channels, in_channel_pos=3, dim_pos_feat=16, aggr="max"):
        super().__init__()
        self.edge_conv_tpl = EdgeConvMotion(nn_x=MLP([in_channels * 2, out_channels // 2, out_channels // 2]),
                                            nn_pos=MLP([in_channel_pos * 2, dim_pos_feat, dim_pos_feat]), aggr=aggr)
        self.edge_conv_geo = EdgeConvMotion(nn_x=MLP([in_channels * 2, out_channels // 2, out_channels // 2]),
                                            nn_pos=MLP([in_channel_pos * 2, dim_pos_feat, dim_pos_feat]), aggr=aggr)
        self.mlp = MLP([out_channels + dim_pos_feat * 2, out_channels])

    def _pack(self):
        vx, (xt, xg) = packing.pack_edge_pair([self.edge_conv_tpl.nn_x, self.edge_conv_geo.nn_x])
        vp, (pt, pg) = packing.pack_edge_pair([self.edge_conv_tpl.nn_pos, self.edge_conv_geo.nn_pos])
        return dict(vx=vx, xt=xt, xg=xg, vp=vp, pt=pt, pg=pg, mlp=packing.pack_mlp_layer(self.mlp[0]))

    def run(self, ops, pos: Mat, x: Mat, csr_tpl, csr_geo, out: Mat, replicas: int = 1, split: bool = False,
            split_in=None, split_out=None):
        """pos: [n, P] window; x: [R*n, C] window (replica-major); out: [R*n, O] window.
        split: x and out are windows in the split-fp16 activation layout (GEMM -> GEMM hand-off);
        split_in / split_out override it for one side."""
        split_in = split if split_in is None else split_in
        split_out = split if split_out is None else split_out
        dev = x.base.device
        pk = self.packed(dev)
        n, M = pos.rows, x.rows
        assert M == n * replicas
        H, D = pk["xt"].H, pk["pt"].H
        ldo = 2 * H + 2 * D
        ab = ops.empty(M, 4 * H, dev)
        ops.gemm(x, pk["vx"], relu=False, Y=Mat.of(ab), x_split=split_in)
        pab = ops.empty(n, 4 * D, dev)
        ops.gemm(pos, pk["vp"], relu=False, Y=Mat.of(pab))
        ec = ops.empty(M, ldo, dev)          # [x_tpl(H) | pos_tpl(D) | x_geo(H) | pos_geo(D)] = torch.cat order (:216)
        ops.edgeconv(Mat.of(ab, 0, H), Mat.of(ab, H, H), csr_tpl, pk["xt"], Mat.of(ec, 0, H),
                     replicas=replicas, in_rep_stride=n, out_rep_stride=n)
        ops.edgeconv(Mat.of(ab, 2 * H, H), Mat.of(ab, 3 * H, H), csr_geo, pk["xg"], Mat.of(ec, H + D, H),
                     replicas=replicas, in_rep_stride=n, out_rep_stride=n)
        # position branch: independent of the keyframe -> computed once into replica 0, copied to the others
        ops.edgeconv(Mat.of(pab, 0, D), Mat.of(pab, D, D), csr_tpl, pk["pt"], Mat.of(ec, H, D, 0, n))
        ops.edgeconv(Mat.of(pab, 2 * D, D), Mat.of(pab, 3 * D, D), csr_geo, pk["pg"], Mat.of(ec, 2 * H + D, D, 0, n))
        if replicas > 1:                                # one launch per column window instead of one per replica
            ops.copy2d_rep(Mat.of(ec, H, D, 0, n), Mat.of(ec, H, D, n, n), replicas - 1, n)
            ops.copy2d_rep(Mat.of(ec, 2 * H + D, D, 0, n), Mat.of(ec, 2 * H + D, D, n, n), replicas - 1, n)
        ops.gemm(Mat.of(ec), pk["mlp"], relu=True, Y=out, y_split=split_out)

    def _forward(self, pos, x, tpl_edge_index, geo_edge_index):
        ops = get_ops()
        x = _padded_copy(ops, _as_matrix(x))
        pos = _padded_copy(ops, _as_matrix(pos))
        n = x.shape[0]
        pk = self.packed(x.device)
        out = ops.empty(n, pk["mlp"].N, x.device)
        self.run(ops, Mat.of(pos, 0, pk["vp"].K), Mat.of(x, 0, pk["vx"].K),
                 ops.csr_build(tpl_edge_index, n), ops.csr_build(geo_edge_index, n), Mat.of(out))
        return out


# ------------------------------------------------------------------------------------------------
# PointNet++ blocks of the CorrNet point branch (models/basic_modules.py:66-138). These classes hold
# the parameters under the reference's state_dict keys; the arithmetic is driven by CorrNet.forward
# (morig_amd/models/corrnet.py) through the op layer.
# ------------------------------------------------------------------------------------------------
class PointConv(torch.nn.Module):
    """parameter holder with PyG PointConv's attribute names (``local_nn``; ``global_nn`` unused)."""

    def __init__(self, local_nn=None, global_nn=None):
        super().__init__()
        self.local_nn = local_nn
        self.global_nn = global_nn


class SAModule(NativeModule):
    """models/basic_modules.py:66-86: fps -> radius ball (<= max_num_neighbors) -> PointConv(max)."""

    def __init__(self, ratio, r, nn, max_num_neighbors):
        super().__init__()
        self.ratio = ratio
        self.r = r
        self.max_num_neighbors = max_num_neighbors
        self.conv = PointConv(nn)

    def _pack(self):
        cx = self.conv.local_nn[0][0].weight.shape[1] - 3
        return packing.pack_pointconv(self.conv.local_nn, cx)


class GlobalSAModule(NativeModule):
    """models/basic_modules.py:115-125: nn([x ‖ pos]) -> global_max_pool."""

    def __init__(self, nn):
        super().__init__()
        self.nn = nn

    def _pack(self):
        return [packing.pack_mlp_layer(l) for l in self.nn]


class FPModule(NativeModule):
    """models/basic_modules.py:127-138: knn_interpolate(k) -> cat skip -> nn."""

    def __init__(self, k, nn):
        super().__init__()
        self.k = k
        self.nn = nn

    def _pack(self):
        return [packing.pack_mlp_layer(l) for l in self.nn]


__all__ += ["PointConv", "SAModule", "GlobalSAModule", "FPModule"]
