"""Drop-in mirror of the reference's ``models/basic_modules.py`` graph blocks: same class names,
constructor signatures, ``forward()`` signatures and ``state_dict`` keys
(/root/reference/models/basic_modules.py:31-36, 142-219); the eval-mode arithmetic runs in the HIP
kernels of libmorig_hip.so through morig_amd.native (no PyG / torch_scatter / torch_cluster).

The torch containers built by ``MLP`` only HOLD parameters (so checkpoints load unchanged); they
are never called on the native path.
"""
from __future__ import annotations

import itertools
import math
import os
import threading
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


class _ForwardContext(threading.local):
    """key of the outermost NativeModule's forward in flight on this thread: nested packed() calls reuse it instead of
    each walking its own parameter subtree (one traversal of ~450 tensors per forward, ~1 ms of host time)."""
    key = None
    root = None
    shift = 0          # range shift k of the forward in flight (see NativeModule.forward): homogeneous stacks run at 2^-k


_ctx = _ForwardContext()
_LAZY_KEY = object()


class NativeModule(torch.nn.Module):
    """Base class: caches packed (kernel-layout) parameters. The cache is keyed on the storage address and the
    autograd version counter of every parameter and buffer below this module, so ANY in-place edit
    (``w.mul_(2)``, an optimizer step), ``load_state_dict`` on a plain ``Sequential`` child, or a reload of a
    child NativeModule whose tensors a parent packs (GCUMotion packs its EdgeConvMotions' MLPs) repacks on
    the next forward; ``load_state_dict`` / ``.to()`` / ``train()`` additionally drop it eagerly.
    NOT seen: edits made through ``.data`` (``p.data.copy_(ema)``, ``p.data.clamp_()``) -- they bypass the version counter;
    call ``invalidate_packed()`` after such an edit."""

    # Range shift [r06]. _RANGE_SCALED: this module's packed constants belong to a positively homogeneous stack (packing.scale_additive) --
    # GCNRig, GCUMotion, SkinNet_inner, the position groups of the rig networks; _RANGE_SHIFT_ROOT: this network's plan scales the inputs of
    # those stacks by 2^-k and their outputs by 2^k (rignet.py), so its forward may answer a split-fp16 range overflow by raising k
    # instead of re-running on the exact path.
    _RANGE_SCALED = False
    _RANGE_SHIFT_ROOT = False
    # (small steps: the smallest sufficient k keeps the most of fp16's narrow exponent range for the ordinary activations -- at 2^-12 values of
    # O(1) would reach fp16's subnormals; the search runs on the first overflowing forward only, k is sticky)
    RANGE_SHIFT_STEP, RANGE_SHIFT_MAX = 2, 10

    def __init__(self):
        super().__init__()
        self._packed = None
        self._packed_device = None
        self._packed_key = None
        self._packed_shift = 0
        self.range_shift = 0               # k, sticky: found by the first forward that overflowed (not part of the state_dict)
        self.register_load_state_dict_post_hook(_invalidate_packed)

    def __getstate__(self):
        """torch.save(model) / copy.deepcopy: the kernel-layout weight cache (device tensors) is derived data"""
        st = super().__getstate__() if hasattr(super(), "__getstate__") else self.__dict__.copy()
        st = dict(st)
        st["_packed"], st["_packed_key"], st["_packed_device"] = None, None, None
        st["_packed_shift"] = 0
        st.pop("_last_key", None)
        return st

    def _param_key(self):
        """(storage, version) of every parameter and buffer below this module. Runs once per forward BEFORE the first launch (the
        GPU is idle meanwhile), so it walks ``_modules`` / ``_parameters`` / ``_buffers`` directly: ``parameters()`` + ``buffers()``
        build a dotted name per tensor and took 0.55-0.76 ms per forward for these networks, this takes ~0.16 ms. An unchanged
        key is returned as the SAME tuple object, so the nested modules' comparisons in ``packed()`` are identity checks."""
        key = []
        stack = [self]
        while stack:
            m = stack.pop()
            if m._modules:
                stack.extend(m._modules.values())
            for d in (m._parameters, m._buffers):
                if d:
                    key.extend((t.data_ptr(), t._version) for t in d.values() if t is not None)
        key = tuple(key)
        last = self.__dict__.get("_last_key")
        if last is not None and last == key:
            return last
        self.__dict__["_last_key"] = key
        return key

    def _drop_packed(self):
        for m in self.modules():
            if isinstance(m, NativeModule):
                m._packed = None

    def invalidate_packed(self):
        """drop the kernel-layout weight cache of this module and everything below it (needed only after parameter edits the
        autograd version counter does not see, i.e. writes through ``.data``)"""
        self._drop_packed()

    def _apply(self, fn, *a, **kw):
        self._drop_packed()
        return super()._apply(fn, *a, **kw)

    def train(self, mode: bool = True):
        self._drop_packed()
        return super().train(mode)

    def packed(self, device):
        key = _ctx.key
        if key is _LAZY_KEY:                           # first use inside this forward: see forward()
            key = _ctx.key = _ctx.root._param_key()
        elif key is None:
            key = self._param_key()
        shift = _ctx.shift if self._RANGE_SCALED else 0
        if self._packed is None or self._packed_device != device or self._packed_key != key or self._packed_shift != shift:
            pk = self._pack()
            if shift:
                pk = packing.scale_additive(pk, 2.0 ** -shift)
            self._packed = packing.to_device(pk, device)
            self._packed_device = device
            self._packed_key = key
            self._packed_shift = shift
        return self._packed

    def _pack(self):
        raise NotImplementedError

    def forward(self, *args, **kwargs):
        """eval-mode forward on the HIP path. The whole forward runs inside the op layer's precision
        guard: split-fp16 MFMA first; if a kernel saw an operand outside the fp16 range the forward is
        repeated with fp32 MFMA (morig_amd.native.NativeOps.guarded)."""
        ops = get_ops()
        dev = next(self.parameters()).device
        if self.training:
            # train mode (batch-statistics BatchNorm, running buffers updated), rignet family only. With autograd enabled the
            # forward is built from the autograd blocks of morig_amd/train_backward.py (native forward AND backward operators:
            # loss.backward() fills every parameter's .grad, as training/train_rig.py:136-195 expects); under no_grad it is the
            # graph-free forward of morig_amd/train_forward.py.
            if type(self)._forward_train is NativeModule._forward_train:
                self._require_eval()
            if torch.is_grad_enabled() and type(self)._forward_train_grad is not NativeModule._forward_train_grad:
                return self._forward_train_grad(*args, **kwargs)
            with torch.no_grad():
                if dev.type == "cuda":
                    with torch.cuda.device(dev):
                        return ops.guarded(dev, lambda: self._forward_train(*args, **kwargs), rerun=False)
                return ops.guarded(dev, lambda: self._forward_train(*args, **kwargs), rerun=False)
        rng = torch.get_rng_state()                    # random FPS starts: a repeated attempt draws the same ones

        def attempt():
            torch.set_rng_state(rng)
            return self._forward(*args, **kwargs)
        outer = _ctx.key
        retry = None
        if outer is None and self._RANGE_SHIFT_ROOT and os.environ.get("MORIG_RANGE_SHIFT", "1") != "0":
            # [r06] an operand beyond the split-fp16 range: before the exact path (2.8 x the time, on EVERY later forward too), run the
            # homogeneous stacks at 2^-k -- inputs and additive constants scaled down, outputs scaled back, all by exact powers of two --
            # and keep k for the forwards that follow (sticky: the weights that overflowed once will again)
            _ctx.shift = self.range_shift

            def retry():
                if self.range_shift >= self.RANGE_SHIFT_MAX:
                    return False
                self.range_shift += self.RANGE_SHIFT_STEP
                _ctx.shift = self.range_shift
                return True
        if outer is None:
            # (storage, version) of every tensor below the outermost module -- computed at the first packed() call of the forward,
            # not here: the plans enqueue their weight-free launches first (CSR builds, sampling), so the ~0.2 ms walk runs while
            # the GPU already works instead of in front of the first launch
            _ctx.key, _ctx.root = _LAZY_KEY, self
        try:
            if dev.type == "cuda":
                # launches go to torch's current stream OF THE MODEL'S DEVICE, whatever the caller's current device is
                with torch.cuda.device(dev):
                    return ops.guarded(dev, attempt, retry=retry)
            return ops.guarded(dev, attempt, retry=retry)
        finally:
            _ctx.key = outer
            if outer is None:
                _ctx.root = None
                _ctx.shift = 0

    def forward_async(self, *args, **kwargs):
        """The eval forward with its guard read DEFERRED: -> (outputs, pending). ``pending.result()`` (morig_amd.native.PendingGuard)
        is the one host read ``forward()`` does before it returns; call it after the NEXT forward has been enqueued and the GPU
        never waits for the host between forwards. result() False = an operand left the split-fp16 range: the outputs are invalid,
        run ``forward()`` (which re-runs on the exact-fp32 kernels). Eval mode, deterministic plans only (no random FPS starts)."""
        assert not self.training, "forward_async is the eval-mode path"
        ops = get_ops()
        dev = next(self.parameters()).device
        outer = _ctx.key
        assert outer is None, "forward_async is for the outermost module"
        _ctx.key, _ctx.root = _LAZY_KEY, self
        _ctx.shift = self.range_shift if self._RANGE_SHIFT_ROOT else 0
        try:
            with torch.cuda.device(dev):
                if not ops.fast:                        # exact-fp32 mode: nothing can overflow, the status words still are checked
                    return ops.guarded(dev, lambda: self._forward(*args, **kwargs)), None
                return ops.guarded_async(dev, lambda: self._forward(*args, **kwargs))
        finally:
            _ctx.key, _ctx.root = outer, None
            _ctx.shift = 0

    def _forward(self, *args, **kwargs):
        raise NotImplementedError

    def _forward_train(self, *args, **kwargs):
        raise NotImplementedError

    def _forward_train_grad(self, *args, **kwargs):
        raise NotImplementedError

    def _require_eval(self):
        if self.training:
            raise NotImplementedError(
                f"{type(self).__name__}: no train-mode forward on the MI355X-native path for this module. The train-mode FORWARD "
                "(batch-statistics BatchNorm over vertices / edges, running-buffer updates) and its BACKWARD pass exist for the "
                "networks the reference trains -- jointnet_motion, masknet_motion, skinnet_motion (morig_amd/train_forward.py, "
                "train_backward.py), corrnet and deformnet (morig_amd/train_corr.py); standalone blocks have neither -- SURVEY.md 8(f-4), DESIGN.md section 9. "
                "Call model.eval() for inference.")


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


def _edge_rows_can_split(ops, ab, ec, csr_tpl, csr_geo, et, eg, H, replicas, n) -> bool:
    """May the two feature EdgeConvs of a unit -- operands [A_tpl | B_tpl | A_geo | B_geo] in `ab`, results into columns [0, H) and
    [H, 2H) of `ec` -- store split-fp16 rows? The library decides (kernel choice, alignment, its environment switches:
    morig_edgeconv_can_split_out); MORIG_EC_SPLIT=0 keeps the fp32 rows (A/B runs)."""
    if not ops.split_activations or not hasattr(ops, "edgeconv_can_split_out") or os.environ.get("MORIG_EC_SPLIT", "1") == "0":
        return False
    kw = dict(replicas=replicas, in_rep_stride=n if replicas > 1 else 0, out_rep_stride=n if replicas > 1 else 0)
    return (ops.edgeconv_can_split_out(Mat.of(ab, 0, H), Mat.of(ab, H, H), csr_tpl, et, Mat.of(ec, 0, H), **kw) and
            ops.edgeconv_can_split_out(Mat.of(ab, 2 * H, H), Mat.of(ab, 3 * H, H), csr_geo, eg, Mat.of(ec, H, H), **kw))


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

    def run(self, ops, x: Mat, csr_tpl, csr_geo, out: Mat, split_in: bool = False, split_out: bool = False):
        """x: [n, C] window (16-byte aligned rows) -> out: [n, O] window. split_in / split_out: the x / out window is in the
        split-fp16 activation layout (GEMM -> GEMM hand-off between consecutive units and into the wide layers)."""
        pk = self.packed(x.base.device)
        n, H = x.rows, pk["et"].H
        dev = x.base.device
        ab = ops.empty(n, 4 * H, dev)
        ops.gemm(x, pk["vertex"], relu=False, Y=Mat.of(ab), x_split=split_in)
        ec = ops.empty(n, 2 * H, dev)
        # [r05] where the library takes it (wide layers on 4-aligned CSRs), the two EdgeConvs store split-fp16 rows and the unit MLP reads
        # them through the LDS-DMA GEMM (morig_edgeconv out_split); [x_tpl | x_geo] is the reference's order (:176), both chunk-aligned
        sp = H % 32 == 0 and _edge_rows_can_split(ops, ab, ec, csr_tpl, csr_geo, pk["et"], pk["eg"], H, 1, 0)
        # (one call for the pair: the boundary passes of the two launches share launches, native.edgeconv_pair)
        ops.edgeconv_pair(dict(A=Mat.of(ab, 0, H), B=Mat.of(ab, H, H), csr=csr_tpl, ec=pk["et"], out=Mat.of(ec, 0, H), out_split=sp),
                          dict(A=Mat.of(ab, 2 * H, H), B=Mat.of(ab, 3 * H, H), csr=csr_geo, ec=pk["eg"], out=Mat.of(ec, H, H), out_split=sp))
        ops.gemm(Mat.of(ec), pk["mlp"], relu=True, Y=out, x_split=sp, y_split=split_out)

    def _forward(self, pos, tpl_edge_index, geo_edge_index):
        ops = get_ops()
        x = _padded_copy(ops, _as_matrix(pos))
        n = x.shape[0]
        pk = self.packed(x.device)
        out = ops.empty(n, pk["mlp"].N, x.device)
        self.run(ops, Mat.of(x, 0, pk["vertex"].K), ops.csr_build(tpl_edge_index, n), ops.csr_build(geo_edge_index, n), Mat.of(out))
        return out


def range_scale() -> float:
    """2^-k of the forward in flight (1.0 outside a range-shifted forward): what the plans of the homogeneous stacks multiply their
    INPUTS by; their outputs are multiplied by the inverse"""
    return 2.0 ** -_ctx.shift


class GCUMotion(NativeModule):
    """models/basic_modules.py:205-219. ``run`` supports R keyframe replicas of the feature with ONE
    position branch (nn_pos depends only on pos and the graph: models/rignet.py:85-86)."""

    _RANGE_SCALED = True

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
        pk = dict(vx=vx, xt=xt, xg=xg, vp=vp, pt=pt, pg=pg, mlp=packing.pack_mlp_layer(self.mlp[0]))
        # [r05] the same layer for the unit's row laid out [x_tpl(H) | x_geo(H) | pos_tpl(D) | pos_geo(D)]: every block then starts on
        # a 32-column chunk, so the wide EdgeConv kernels can store split-fp16 rows (morig_edgeconv out_split) and this GEMM reads them
        # through the LDS-DMA kernel (reference order, :216: [x_tpl | pos_tpl | x_geo | pos_geo])
        H, D = xt.H, pt.H
        if H % 32 == 0 and (2 * D) % 32 == 0:
            cols = (list(range(H)) + [2 * H + i for i in range(D)] + [H + i for i in range(H)] + [2 * H + D + i for i in range(D)])
            pk["mlp_s"] = packing.pack_mlp_layer(self.mlp[0], in_cols=cols, k_total=2 * H + 2 * D)
        # a 3-channel feature with a 32-wide hidden layer (motionNet's first unit: the keyframe flow): the first Linear in the form
        # morig_edgeconv_x3 evaluates in its loader
        if xt.H == 32 and self.edge_conv_tpl.nn_x[0][0].weight.shape[1] == 6:
            firsts = []
            for ec in (self.edge_conv_tpl, self.edge_conv_geo):
                W1 = ec.nn_x[0][0].weight.detach().float()
                firsts.append(packing.pack_first_x3(W1[:, :3] - W1[:, 3:], W1[:, 3:], ec.nn_x[0][0].bias.detach().float()))
            pk["x3t"], pk["x3g"] = firsts
        return pk

    def run(self, ops, pos: Mat, x: Mat, csr_tpl, csr_geo, out: Mat, replicas: int = 1, split: bool = False,
            split_in=None, split_out=None, pos_feat=None, x3: Optional[Mat] = None):
        """pos: [n, P] window; x: [R*n, C] window (replica-major); out: [R*n, O] window.
        split: x and out are windows in the split-fp16 activation layout (GEMM -> GEMM hand-off);
        split_in / split_out override it for one side.
        pos_feat: (Mat [n, D], Mat [n, D][, Mat [n, 32] or None]) = this unit's position-branch results on the tpl / geo graph, already
        computed by ``run_pos_groups`` (paired with another unit's): copied into every replica instead of being computed here. The third
        entry, when given, is the same pair once more as ONE split-fp16 chunk per vertex [pos_tpl | pos_geo] (ops.pack_tails): where the
        unit's MLP runs on the LDS-DMA store kernel it reads that block as the K TAIL of its input (row v of the tail for the rows
        r n + v of all replicas: morig_gemm_args.X_tail) and nothing is copied at all.
        x3: the 3-channel feature as plain fp32 rows [R*n, 4] (first unit of motionNet): its two EdgeConvs then evaluate the first
        Linear in-kernel from the gathered endpoints (morig_edgeconv_x3) and the [A | B] GEMM is not run."""
        split_in = split if split_in is None else split_in
        split_out = split if split_out is None else split_out
        dev = x.base.device
        pk = self.packed(dev)
        n, M = pos.rows, x.rows
        assert M == n * replicas
        H, D = pk["xt"].H, pk["pt"].H
        ldo = 2 * H + 2 * D
        use_x3 = x3 is not None and "x3t" in pk and hasattr(ops, "edgeconv_x3") and os.environ.get("MORIG_EDGE_X3", "1") != "0"
        if not use_x3:
            ab = ops.empty(M, 4 * H, dev)
            ops.gemm(x, pk["vx"], relu=False, Y=Mat.of(ab), x_split=split_in)
        if pos_feat is None:
            pab = ops.empty(n, 4 * D, dev)
            ops.gemm(pos, pk["vp"], relu=False, Y=Mat.of(pab))
        tail = pos_feat[2] if (pos_feat is not None and len(pos_feat) > 2) else None
        use_tail = (tail is not None and not use_x3 and "mlp_s" in pk and hasattr(ops, "gemm_takes_tail")
                    and ops.gemm_takes_tail(pk["mlp_s"], out, 2 * D))
        ec = ops.empty(M, ldo, dev)          # [x_tpl(H) | pos_tpl(D) | x_geo(H) | pos_geo(D)] = torch.cat order (:216)
        if not use_x3 and "mlp_s" in pk and _edge_rows_can_split(ops, ab, ec, csr_tpl, csr_geo, pk["xt"], pk["xg"], H, replicas, n):
            # split-fp16 rows [x_tpl | x_geo | pos_tpl | pos_geo] straight from the EdgeConv kernels into the LDS-DMA GEMM
            ops.edgeconv_pair(dict(A=Mat.of(ab, 0, H), B=Mat.of(ab, H, H), csr=csr_tpl, ec=pk["xt"], out=Mat.of(ec, 0, H),
                                   replicas=replicas, in_rep_stride=n, out_rep_stride=n, out_split=True),
                              dict(A=Mat.of(ab, 2 * H, H), B=Mat.of(ab, 3 * H, H), csr=csr_geo, ec=pk["xg"], out=Mat.of(ec, H, H),
                                   replicas=replicas, in_rep_stride=n, out_rep_stride=n, out_split=True))
            if use_tail:
                # [r06] the replica-invariant block is not copied into the replicas' rows: the GEMM reads it from the one tail row per vertex
                # (the pos columns of `ec` stay unwritten and unread)
                ops.gemm(Mat.of(ec, 0, 2 * H), pk["mlp_s"], relu=True, Y=out, x_split=True, y_split=split_out, x_tail=tail)
                return
            p2 = ops.empty(n, 2 * D, dev)               # [pos_tpl | pos_geo] side by side: ONE chunk-aligned split copy into every replica
            if pos_feat is not None:
                # (library copies, not torch.cat: with strided inputs torch's cat did not replay from a captured HIP graph --
                # tests/test_gpu_networks.py::test_captured_forward_replays_bit_identically)
                ops.copy2d(pos_feat[0], Mat.of(p2, 0, D))
                ops.copy2d(pos_feat[1], Mat.of(p2, D, D))
            else:
                ops.edgeconv_pair(dict(A=Mat.of(pab, 0, D), B=Mat.of(pab, D, D), csr=csr_tpl, ec=pk["pt"], out=Mat.of(p2, 0, D)),
                                  dict(A=Mat.of(pab, 2 * D, D), B=Mat.of(pab, 3 * D, D), csr=csr_geo, ec=pk["pg"], out=Mat.of(p2, D, D)))
            ops.copy2d_rep(Mat.of(p2), Mat.of(ec, 2 * H, 2 * D, 0, n), replicas, n, split=True)
            ops.gemm(Mat.of(ec), pk["mlp_s"], relu=True, Y=out, x_split=True, y_split=split_out)
            return
        if use_x3:
            ops.edgeconv_x3_pair(dict(X=x3, first=pk["x3t"], csr=csr_tpl, ec=pk["xt"], out=Mat.of(ec, 0, H),
                                      replicas=replicas, in_rep_stride=n, out_rep_stride=n),
                                 dict(X=x3, first=pk["x3g"], csr=csr_geo, ec=pk["xg"], out=Mat.of(ec, H + D, H),
                                      replicas=replicas, in_rep_stride=n, out_rep_stride=n))
        else:
            ops.edgeconv_pair(dict(A=Mat.of(ab, 0, H), B=Mat.of(ab, H, H), csr=csr_tpl, ec=pk["xt"], out=Mat.of(ec, 0, H),
                                   replicas=replicas, in_rep_stride=n, out_rep_stride=n),
                              dict(A=Mat.of(ab, 2 * H, H), B=Mat.of(ab, 3 * H, H), csr=csr_geo, ec=pk["xg"], out=Mat.of(ec, H + D, H),
                                   replicas=replicas, in_rep_stride=n, out_rep_stride=n))
        # position branch: independent of the keyframe -> computed once, copied to every replica
        if pos_feat is not None:
            ops.copy2d_rep(pos_feat[0], Mat.of(ec, H, D, 0, n), replicas, n)
            ops.copy2d_rep(pos_feat[1], Mat.of(ec, 2 * H + D, D, 0, n), replicas, n)
        else:
            ops.edgeconv_pair(dict(A=Mat.of(pab, 0, D), B=Mat.of(pab, D, D), csr=csr_tpl, ec=pk["pt"], out=Mat.of(ec, H, D, 0, n)),
                              dict(A=Mat.of(pab, 2 * D, D), B=Mat.of(pab, 3 * D, D), csr=csr_geo, ec=pk["pg"], out=Mat.of(ec, 2 * H + D, D, 0, n)))
            if replicas > 1:                            # one launch per column window instead of one per replica
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


def run_pos_groups(ops, packed_groups, pos: Mat, csr_tpl, csr_geo):
    """The position branches of several GCUMotion units in pairs (packing.pack_pos_groups): per pair one 32-wide EdgeConv per graph.
    With 3-channel positions the first Linear is evaluated inside the EdgeConv kernel from the gathered endpoints
    (morig_edgeconv_x3: no per-vertex [A | B] table, 32 instead of 256 gathered bytes per edge row); wider position inputs (SkinNet's
    33 channels are not paired: D = 64) would take one vertex GEMM for all pairs.
    -> per covered unit (Mat tpl [n, D], Mat geo [n, D], tail): the first two are windows of one side buffer; tail = the pair once more
    as one split-fp16 chunk per vertex, [n, 32] = [pos_tpl | pos_geo | 0] (ops.pack_tails: ONE launch for all units; None off the
    split-fp16 path or for D > 16), which the units' MLPs read as the K tail of their input (GCUMotion.run)."""
    vertex, edges, n_pairs = packed_groups
    if n_pairs == 0:
        return []
    dev = pos.base.device
    n = pos.rows
    H = edges[0][0].H                               # 2 D
    D = H // 2
    x3 = pos.cols == 3 and pos.col0 == 0 and pos.ld % 4 == 0 and all(e[2] is not None for e in edges) and hasattr(ops, "edgeconv_x3") \
        and os.environ.get("MORIG_EDGE_X3", "1") != "0"
    side = ops.empty(n, 2 * H * n_pairs, dev)       # per pair [tpl: unit 0, unit 1 | geo: unit 0, unit 1]
    if not x3:
        pab = ops.empty(n, 4 * H * n_pairs, dev)    # per pair [A_tpl | B_tpl | A_geo | B_geo], H columns each
        ops.gemm(pos, vertex, relu=False, Y=Mat.of(pab))
    out = []
    for g, (et, eg, ft, fg) in enumerate(edges):
        if x3:
            ops.edgeconv_x3_pair(dict(X=pos, first=ft, csr=csr_tpl, ec=et, out=Mat.of(side, 2 * H * g, H)),
                                 dict(X=pos, first=fg, csr=csr_geo, ec=eg, out=Mat.of(side, 2 * H * g + H, H)))
        else:
            c0 = 4 * H * g
            ops.edgeconv_pair(dict(A=Mat.of(pab, c0, H), B=Mat.of(pab, c0 + H, H), csr=csr_tpl, ec=et, out=Mat.of(side, 2 * H * g, H)),
                              dict(A=Mat.of(pab, c0 + 2 * H, H), B=Mat.of(pab, c0 + 3 * H, H), csr=csr_geo, ec=eg, out=Mat.of(side, 2 * H * g + H, H)))
        for j in range(2):
            out.append((Mat.of(side, 2 * H * g + j * D, D), Mat.of(side, 2 * H * g + H + j * D, D)))
    if ops.split_activations and 2 * D <= 32 and len(out) <= 8 and hasattr(ops, "pack_tails") and os.environ.get("MORIG_GEMM_TAIL", "1") != "0":
        tails = ops.pack_tails(side, [t.col0 for t, _ in out], [g_.col0 for _, g_ in out], D, D)
        out = [(t, g_, Mat.of(tails[u])) for u, (t, g_) in enumerate(out)]
    else:
        out = [(t, g_, None) for t, g_ in out]
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


def radius_cpu(x, y, r, max_num_neighbors):
    """models/basic_modules.py:9-29 -- the ball query of the reference's no-CUDA branch, same signature and result layout:
    ``[x_idx ; y_idx]`` (2 x E, int64) of all pairs with dist(y, x) <= r (inclusive, ``batch`` ignored); rows of y with at
    most ``max_num_neighbors`` hits first, row-major with x indices ascending, then the over-full rows with exactly
    ``max_num_neighbors`` hits each. The reference draws those with torch.multinomial on the 0/1 validity row (uniform over
    subsets); here the HIP kernel keeps a uniform reservoir sample seeded from torch's RNG -- the same distribution, not the
    same stream (SURVEY.md section 7 hard part 4). Runs on the device the points live on."""
    ops = get_ops()
    xm = _pos4(ops, x)
    ym = _pos4(ops, y)
    seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
    coo, counts = ops.radius_sample(Mat.of(xm, 0, 3), Mat.of(ym, 0, 3), r, max_num_neighbors, seed)
    ny = ym.shape[0]
    slots = coo[0].view(ny, max_num_neighbors)
    rows = torch.arange(ny, device=slots.device).view(-1, 1).expand_as(slots)
    reduced = (counts > max_num_neighbors).view(-1, 1)
    keep_res = (slots >= 0) & ~reduced
    keep_red = reduced.expand_as(slots)
    return torch.cat([torch.stack([slots[keep_res], rows[keep_res]]), torch.stack([slots[keep_red], rows[keep_red]])], dim=1)


def _cloud_offsets(batch: torch.Tensor, n_clouds: Optional[int] = None):
    """sorted PyG ``batch`` vector -> (per-cloud counts as a host list, int32 offsets on the device). One host sync,
    as torch_cluster's own ``batch`` handling has (it needs the cloud sizes to size its outputs)."""
    counts = torch.bincount(batch, minlength=n_clouds or 0).tolist()
    off = [0]
    for c in counts:
        off.append(off[-1] + int(c))
    return counts, torch.tensor(off, dtype=torch.int32, device=batch.device)


def _with_pos(ops, x: Optional[torch.Tensor], pos4: torch.Tensor):
    """[x | pos | 0] with a 16-byte aligned row stride (x may be None); -> (buffer, width of x)."""
    if x is None:
        return pos4, 0
    n, c = x.shape
    ld = (c + 3 + 3) // 4 * 4
    buf = torch.zeros((n, ld), dtype=torch.float32, device=x.device)
    ops.copy2d(Mat.of(x), Mat.of(buf, 0, c))
    ops.copy2d(Mat.of(pos4, 0, 3), Mat.of(buf, c, 3))
    return buf, c


def _pos4(ops, pos: torch.Tensor) -> torch.Tensor:
    p4 = torch.zeros((pos.shape[0], 4), dtype=torch.float32, device=pos.device)
    ops.copy2d(Mat.of(pos.float().contiguous()), Mat.of(p4, 0, 3))
    return p4


class SAModule(NativeModule):
    """models/basic_modules.py:66-86: fps -> radius ball (<= max_num_neighbors) -> PointConv(max).
    Semantics of the branch the reference takes on a GPU (deterministic ``radius``: first hits in index order,
    strict <); ``radius_cpu`` (:9-29) draws a random subset with torch.multinomial and ignores ``batch``."""

    def __init__(self, ratio, r, nn, max_num_neighbors):
        super().__init__()
        self.ratio = ratio
        self.r = r
        self.max_num_neighbors = max_num_neighbors
        self.conv = PointConv(nn)

    def _pack(self):
        cx = self.conv.local_nn[0][0].weight.shape[1] - 3
        return packing.pack_pointconv(self.conv.local_nn, cx)

    def neighbours(self, ops, xp: torch.Tensor, cx: int, pos_new: torch.Tensor, ptr: torch.Tensor, out_ptr: torch.Tensor, n_clouds: int):
        """the ball query of ``run`` (slot table): positions only, so a caller can issue it before the features exist"""
        return ops.ball_query(Mat.of(xp, cx, 3), ptr, Mat.of(pos_new, 0, 3), out_ptr, n_clouds, self.r, self.max_num_neighbors)

    def run(self, ops, xp: torch.Tensor, cx: int, pos_new: torch.Tensor, ptr: torch.Tensor, out_ptr: torch.Tensor, n_clouds: int,
            coo: Optional[torch.Tensor] = None):
        """xp: [N, ld] = [x(cx) | pos(3) | pad]; pos_new: the sampled centres [M, 4]; ptr / out_ptr: int32 cloud offsets
        of the sources / centres; returns x_new [M, H3].
        PointConv's first Linear on [x_j ‖ pos_j - pos_i] splits per point: B_j = W1 [x_j ‖ pos_j] + b1, A_i = -W1p pos_i."""
        dev = xp.device
        pk = self.packed(dev)
        N, M = xp.shape[0], pos_new.shape[0]
        if coo is None:
            coo = self.neighbours(ops, xp, cx, pos_new, ptr, out_ptr, n_clouds)
        H = pk["edge"].H
        bsrc = ops.empty(N, H, dev)
        ops.gemm(Mat.of(xp, 0, cx + 3), pk["src"], relu=False, Y=Mat.of(bsrc))
        atgt = ops.empty(M, H, dev)
        ops.gemm(Mat.of(pos_new, 0, 3), pk["tgt"], relu=False, Y=Mat.of(atgt))
        x_new = ops.empty(M, pk["last"].N, dev)
        if ops.pointconv_can_fuse(pk, self.max_num_neighbors):
            # layers 2 and 3 and the max in one kernel, straight from the slot table: no CSR, no per-edge rows in HBM
            ops.pointconv_fused(Mat.of(atgt), Mat.of(bsrc), coo, self.max_num_neighbors, pk, Mat.of(x_new))
            return x_new
        csr = ops.csr_from_slots(coo, M, self.max_num_neighbors, N)
        z = ops.empty(csr.capacity, H, dev)
        ops.edge_hidden(Mat.of(atgt), Mat.of(bsrc), csr, pk["edge"], Mat.of(z))
        ops.segmax_gemm(Mat.of(z), pk["last"], True, csr, Mat.of(x_new))
        return x_new

    def _forward(self, x, pos, batch, random_start=True):
        """-> (x_new [M, H3], pos[idx] [M, 3], batch[idx] [M])   (models/basic_modules.py:74-86)"""
        ops = get_ops()
        dev = pos.device
        counts, ptr = _cloud_offsets(batch)
        B = len(counts)
        out_counts = [int(math.ceil(self.ratio * c)) for c in counts]
        _, out_ptr = _cloud_offsets(torch.repeat_interleave(torch.arange(B, device=dev), torch.tensor(out_counts, device=dev)), B)
        start = None
        if random_start:                                  # same draw order as torch_cluster.fps: one randint per cloud
            start = torch.tensor([int(torch.randint(c, (1,))) for c in counts], dtype=torch.int32, device=dev)
        p4 = _pos4(ops, pos)
        M = sum(out_counts)
        idx = ops.fps(Mat.of(p4, 0, 3), ptr, out_ptr, start, B, max(counts), M)
        pos_new = torch.zeros((M, 4), dtype=torch.float32, device=dev)
        ops.gather_rows(Mat.of(p4, 0, 3), idx, Mat.of(pos_new, 0, 3))
        xp, cx = _with_pos(ops, None if x is None else x.float().contiguous(), p4)
        x_new = self.run(ops, xp, cx, pos_new, ptr, out_ptr, B)
        return x_new, pos_new[:, :3].contiguous(), batch[idx.long()]


class GlobalSAModule(NativeModule):
    """models/basic_modules.py:115-125: nn([x ‖ pos]) -> global_max_pool."""

    def __init__(self, nn):
        super().__init__()
        self.nn = nn

    def _pack(self):
        return [packing.pack_mlp_layer(l) for l in self.nn]

    def run(self, ops, xp: torch.Tensor, width: int, seg: torch.Tensor, n_clouds: int) -> torch.Tensor:
        """xp: [M, ld] = [x | pos | pad], ``width`` = columns of [x | pos]; seg: int32 cloud id per row (sorted);
        -> pooled [n_clouds, C_out]: the last layer's per-cloud column max is the epilogue of its GEMM."""
        dev = xp.device
        layers = self.packed(dev)
        h = Mat.of(xp, 0, width)
        for lay in layers[:-1]:
            o = ops.empty(xp.shape[0], lay.N, dev)
            ops.gemm(h, lay, relu=True, Y=Mat.of(o))
            h = Mat.of(o)
        pooled = ops.empty(n_clouds, layers[-1].N, dev)
        ops.gemm(h, layers[-1], relu=True, seg=seg, pool=pooled)
        return pooled

    def _forward(self, x, pos, batch):
        """-> (pooled [B, C_out], zeros [B, 3], arange(B))   (models/basic_modules.py:121-125)"""
        ops = get_ops()
        dev = pos.device
        B = int(batch.max().item()) + 1
        xp, cx = _with_pos(ops, x.float().contiguous(), _pos4(ops, pos))
        pooled = self.run(ops, xp, cx + 3, ops.make_seg(batch, B, 1), B)
        return pooled, pos.new_zeros((B, 3)), torch.arange(B, device=dev)


class FPModule(NativeModule):
    """models/basic_modules.py:127-138: knn_interpolate(k) -> cat skip -> nn."""

    def __init__(self, k, nn):
        super().__init__()
        self.k = k
        self.nn = nn

    def _pack(self):
        return [packing.pack_mlp_layer(l) for l in self.nn]

    def search(self, ops, pos_x: torch.Tensor, ptr_x: torch.Tensor, pos_y: torch.Tensor, ptr_y: torch.Tensor, n_clouds: int,
               max_targets_per_cloud: int):
        """the geometry half of ``run`` (nearest sources and weights): depends on positions only, so a caller can issue it early"""
        return ops.knn_search(Mat.of(pos_x, 0, 3), ptr_x, Mat.of(pos_y, 0, 3), ptr_y, n_clouds, max_targets_per_cloud, self.k)

    def run(self, ops, feat: torch.Tensor, pos_x: torch.Tensor, ptr_x: torch.Tensor, skip: Optional[torch.Tensor],
            pos_y: torch.Tensor, ptr_y: torch.Tensor, n_clouds: int, max_targets_per_cloud: int, nn=None) -> torch.Tensor:
        """feat [Nx, Cf] at pos_x ([Nx, >=3]) interpolated onto pos_y ([Ny, >=3]), concatenated with skip [Ny, Cs], then the MLP.
        nn: the result of ``search`` for the same geometry (None: searched here)."""
        dev = feat.device
        layers = self.packed(dev)
        ny, cf = pos_y.shape[0], feat.shape[1]
        cs = 0 if skip is None else skip.shape[1]
        ld = (cf + cs + 3) // 4 * 4                       # GEMM operand rows are 16-byte aligned; the padding stays zero
        cat = ops.empty(ny, ld, dev) if ld == cf + cs else torch.zeros((ny, ld), dtype=torch.float32, device=dev)
        if nn is None:
            nn = self.search(ops, pos_x, ptr_x, pos_y, ptr_y, n_clouds, max_targets_per_cloud)
        ops.knn_apply(Mat.of(feat), nn, Mat.of(cat, 0, cf))
        if skip is not None:
            ops.copy2d(Mat.of(skip), Mat.of(cat, cf, cs))
        h = Mat.of(cat, 0, cf + cs)
        for lay in layers:
            o = ops.empty(ny, lay.N, dev)
            ops.gemm(h, lay, relu=True, Y=Mat.of(o))
            h = Mat.of(o)
        return h.base

    def _forward(self, x, pos, batch, x_skip, pos_skip, batch_skip):
        """-> (nn([interp ‖ x_skip]), pos_skip, batch_skip)   (models/basic_modules.py:133-138)"""
        ops = get_ops()
        B = int(max(int(batch.max().item()), int(batch_skip.max().item()))) + 1
        _, ptr_x = _cloud_offsets(batch, B)
        cy, ptr_y = _cloud_offsets(batch_skip, B)
        out = self.run(ops, x.float().contiguous(), _pos4(ops, pos), ptr_x, None if x_skip is None else x_skip.float().contiguous(),
                       _pos4(ops, pos_skip), ptr_y, B, max(cy))
        return out, pos_skip, batch_skip


__all__ += ["PointConv", "SAModule", "GlobalSAModule", "FPModule", "radius_cpu"]
