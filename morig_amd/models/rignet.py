"""Drop-in mirror of the reference's ``models/rignet.py`` (same names, signatures, state_dict keys;
/root/reference/models/rignet.py:10-220) running eval-mode forwards on the MI355X-native op layer.

Data-flow restructurings (all exact up to fp32 rounding):
  * the ``num_keyframes`` loop over ``motionNet`` (:85-88) is ONE pass over R = num_keyframes
    replicas of the vertex set (rows r*V + v); the position branch of every EdgeConvMotion is
    computed once and shared by the replicas;
  * ``torch.cat([...])`` inputs (:62, :65) are column windows of one wide activation buffer
    ``[x_1 | x_2 | x_3 | pos,0 | feature]``; the consuming Linear's weight columns are permuted once;
  * ``scatter_max`` + ``repeat_interleave`` (:63-64) = pooled epilogue of the mlp_glb GEMM + a
    per-mesh row bias (x_global @ W_g^T) in the first mlp_transform layer;
  * TemporalAttn (:36-46) keeps only token 0, so only the CLS query is evaluated.
"""
from __future__ import annotations

import math
import os
from typing import Optional

import numpy as np
import torch
from torch.nn import Linear, Sequential

from .. import packing
from ..native import Mat
from ..runtime import get_ops
from .basic_modules import MLP, GCUMotion, NativeModule, _padded_copy, range_scale, run_pos_groups

__all__ = ["jointnet_motion", "masknet_motion", "skinnet_motion"]


def _num_graphs(data, batch: torch.Tensor) -> int:
    ng = getattr(data, "num_graphs", None)
    if ng is None:
        ng = int(batch.max().item()) + 1          # one host sync, as torch_scatter's dim_size inference does
    return int(ng)


class TemporalAttn(NativeModule):
    """models/rignet.py:10-46."""

    def __init__(self, input_size, num_heads, hidden_size, dim_feedforward, output_size):
        super().__init__()
        self.num_heads = num_heads
        self.w_qs = Linear(input_size, hidden_size * num_heads, bias=False)
        self.w_ks = Linear(input_size, hidden_size * num_heads, bias=False)
        self.w_vs = Linear(input_size, hidden_size * num_heads, bias=False)
        self.w_o = Linear(hidden_size * num_heads, hidden_size, bias=False)
        self.feedforward = MLP([hidden_size, dim_feedforward, output_size])
        self.cls_token = torch.nn.Parameter(torch.randn(1, 1, input_size))

    def _pack(self):
        nh = self.num_heads
        Wq, Wk, Wv, Wo = (w.weight.detach().double() for w in (self.w_qs, self.w_ks, self.w_vs, self.w_o))
        d = Wq.shape[0] // nh
        cls = self.cls_token.detach().double().reshape(-1)
        q = Wq @ cls                                          # CLS query, all heads
        g, mix = [], []
        for h in range(nh):
            sl = slice(h * d, (h + 1) * d)
            g.append(Wk[sl].t() @ q[sl] / math.sqrt(d))       # score_t = <tok_t, g_h>
            mix.append(Wo[:, sl] @ Wv[sl])                    # w_o(head h of values) = (Wo_h Wv_h) y_h
        return dict(g=torch.stack(g).float().contiguous(), cls=cls.float().contiguous(),
                    mix=packing.pack_linear(torch.cat(mix, dim=1).float()),
                    ff1=packing.pack_mlp_layer(self.feedforward[0]),
                    ff2=packing.pack_mlp_layer(self.feedforward[1]))

    def run(self, ops, x: torch.Tensor, out: Mat):
        """x: [n, T, C] contiguous -> out [n, output_size] window."""
        dev = x.device
        pk = self.packed(dev)
        n = x.shape[0]
        y = ops.empty(n, pk["g"].shape[0] * x.shape[2], dev)
        ops.cls_attention(x, pk["g"], pk["cls"], Mat.of(y))
        res = ops.empty(n, pk["mix"].N, dev)
        ops.gemm(Mat.of(y), pk["mix"], relu=False, Y=Mat.of(res))
        h = ops.empty(n, pk["ff1"].N, dev)
        ops.gemm(Mat.of(res), pk["ff1"], relu=True, Y=Mat.of(h))
        ops.gemm(Mat.of(h), pk["ff2"], relu=True, Y=out)

    def _forward(self, x):
        ops = get_ops()
        x = x.float().contiguous()
        out = ops.empty(x.shape[0], self.feedforward[1][0].out_features, x.device)
        self.run(ops, x, Mat.of(out))
        return out


class GCNRig(NativeModule):
    """models/rignet.py:49-67 (and, through ``WIDTHS`` / ``TRANSFORM``, the identically wired
    ``GCNDeform`` of models/deformnet.py:13-32, see morig_amd/models/deformnet.py)."""

    # wide activation buffer: [x_1 | x_2 | x_3 | pos slot 32 (xyz, zeros) | feature slot roundup32(F)];
    # every window starts on a 32-column chunk so the buffer can be kept in the split-fp16 layout
    WIDTHS = (64, 256, 512)
    TRANSFORM = "mlp_transform"
    _RANGE_SCALED = True          # positively homogeneous in (pos, feature, additive constants): NativeModule's range shift

    def __init__(self, chn_feature, chn_output, aggr="max"):
        super().__init__()
        self.chn_feature, self.chn_output = chn_feature, chn_output
        w1, w2, w3 = self.WIDTHS
        self.gcu_1 = GCUMotion(in_channels=chn_feature, out_channels=w1, dim_pos_feat=16, aggr=aggr)
        self.gcu_2 = GCUMotion(in_channels=w1, out_channels=w2, dim_pos_feat=16, aggr=aggr)
        self.gcu_3 = GCUMotion(in_channels=w2, out_channels=w3, dim_pos_feat=16, aggr=aggr)
        self.mlp_glb = MLP([(w1 + w2 + w3), 1024])
        setattr(self, self.TRANSFORM,
                Sequential(MLP([1024 + 3 + chn_feature + w1 + w2 + w3, 1024, 256]), Linear(256, chn_output)))

    # column offsets inside the wide buffer
    @property
    def X1(self): return 0
    @property
    def X2(self): return self.WIDTHS[0]
    @property
    def X3(self): return self.WIDTHS[0] + self.WIDTHS[1]
    @property
    def POS(self): return sum(self.WIDTHS)
    @property
    def FEAT(self): return sum(self.WIDTHS) + 32

    @property
    def feat_slot(self):
        return (self.chn_feature + 31) // 32 * 32

    @property
    def wide_ld(self):
        return self.FEAT + self.feat_slot

    def _pack(self):
        F = self.chn_feature
        tr = getattr(self, self.TRANSFORM)
        l1 = tr[0][0]
        W = l1[0].weight.detach()
        # reference column order of the transform's input (:65): [x_global(1024) | pos(3) | feature(F) | x_1 x_2 x_3]
        Wg, Wrest = W[:, :1024], W[:, 1024:]
        in_cols = ([self.POS + i for i in range(3)] + [self.FEAT + i for i in range(F)] + list(range(self.POS)))
        t1 = packing.pack_linear(Wrest, l1[0].bias, l1[2], in_cols=in_cols, k_total=self.FEAT + F)
        t1m = (packing.pack_linear(Wrest, l1[0].bias, l1[2],
                                   in_cols=[self.POS + i for i in range(3)] + [self.POS + 4 + i for i in range(F)] + list(range(self.POS)),
                                   k_total=self.POS + 4 + F) if F == 3 else None)
        # t1 and t1m hold the same rows in two column orders, so their row factors agree and ONE row bias `g` serves both
        assert t1m is None or t1.row_factor is None or torch.equal(t1.row_factor, t1m.row_factor)
        return dict(
            glb=packing.pack_mlp_layer(self.mlp_glb[0]),
            g=packing.couple_rowbias(packing.pack_linear(Wg), t1),       # x_global @ Wg^T  -> per-mesh row bias (in t1's row units)
            t1=t1,
            # [r05] merged layout (3-channel feature next to the positions in ONE 32-column chunk: K = POS + 7 -> POS + 32 instead of
            # POS + 35 -> POS + 64, one K chunk of 28 less in the largest GEMM of the motion pass); used when gcu_1 runs on the raw
            # 3-channel rows (morig_edgeconv_x3) and therefore never reads the feature window of the wide buffer (`run`)
            t1m=t1m,
            t2=packing.pack_mlp_layer(tr[0][1]),
            t3=packing.pack_linear(tr[1].weight, tr[1].bias),
        )

    def run(self, ops, pos4: torch.Tensor, write_feature, csr_tpl, csr_geo, seg: torch.Tensor, n_graphs: int,
            replicas: int, out: Mat, csr_geo_wide=None, csr_tpl_wide=None, pos_feats=None, feat3: Optional[Mat] = None,
            posfeat8: Optional[Mat] = None):
        """pos4: [n, 4] (pos, 0); write_feature(window Mat [R*n, feat_slot], split) fills the feature slot
        (zero padded); seg: int32 [R*n] = r*n_graphs + batch[v]; out: [R*n, chn_output] window.
        pos_feats: per unit the position-branch results computed ahead by run_pos_groups (or None entries).
        feat3: a 3-channel feature once more as plain fp32 rows [R*n, 4] (gcu_1 then runs morig_edgeconv_x3 on it).
        posfeat8: [R*n, 8] plain fp32 rows [pos xyz 0 | feature xyz 0] (the caller built them with library copies; feat3 is its
        columns 4..7): the merged 32-column chunk at POS is ONE split copy of it."""
        pf = list(pos_feats) if pos_feats is not None else [None, None, None]
        dev = pos4.device
        pk = self.packed(dev)
        n, R, F = pos4.shape[0], replicas, self.chn_feature
        M = n * R
        sp = ops.split_activations                    # GEMM -> GEMM activations in the split-fp16 layout
        merged = (F == 3 and feat3 is not None and pk.get("t1m") is not None and hasattr(ops, "edgeconv_x3")
                  and os.environ.get("MORIG_EDGE_X3", "1") != "0" and os.environ.get("MORIG_MERGED_POS_FEAT", "1") != "0"
                  and "x3t" in self.gcu_1.packed(dev))
        if merged:
            # [pos xyz 0 | feature 0 | zeros] in the ONE chunk at POS: rows of pos4 repeated per replica beside the plain feature rows
            feat_col, k_t1, t1 = self.POS + 4, self.POS + 4 + F, pk["t1m"]
            wide = ops.empty(M, self.POS + 32, dev)
            if posfeat8 is None:
                # (library copies, no torch.cat / repeat: three ATen launches less, and strided cat inputs did not replay from a captured
                # HIP graph -- ADVICE r5)
                pf8 = ops.empty(M, 8, dev)
                ops.copy2d_rep(Mat.of(pos4, 0, 4), Mat.of(pf8, 0, 4, 0, n), R, n)
                ops.copy2d(Mat.of(feat3.base, feat3.col0, 4, feat3.row0, feat3.rows), Mat.of(pf8, 4, 4))
                posfeat8 = Mat.of(pf8)
            ops.copy2d_pad(posfeat8, Mat.of(wide, self.POS, 32), split=sp)
        else:
            feat_col, k_t1, t1 = self.FEAT, self.FEAT + F, pk["t1"]
            wide = ops.empty(M, self.wide_ld, dev)
            ops.copy2d_rep(Mat.of(pos4), Mat.of(wide, self.POS, 32, 0, n), R, n, split=sp)       # the same positions in every replica
            write_feature(Mat.of(wide, self.FEAT, self.feat_slot), sp)
        posm = Mat.of(pos4, 0, 3)
        self.gcu_1.run(ops, posm, Mat.of(wide, feat_col, F), csr_tpl, csr_geo, Mat.of(wide, self.X1, self.WIDTHS[0]), R, split=sp,
                       pos_feat=pf[0], x3=feat3 if F == 3 else None)
        # the 128- and 256-wide layers take both graphs with 4-aligned segments (quad-reduced, single-pass epilogue of the
        # wave-specialised kernel: +12..18 % on the geo graph; on the tpl graph, in-degree 7 -> 8 rows, still -4 % since the
        # scans became cheap: 54.05 -> 53.35 ms per step); on the narrow layers the padding costs more than it saves
        cg = csr_geo_wide or csr_geo
        ct = csr_tpl_wide or csr_tpl
        self.gcu_2.run(ops, posm, Mat.of(wide, self.X1, self.WIDTHS[0]), ct, cg, Mat.of(wide, self.X2, self.WIDTHS[1]), R, split=sp,
                       pos_feat=pf[1])
        # [r06] MORIG_EDGE_MIX=1 (opt-in): the 256-wide edge layers run WITHOUT padded rows on plain CSRs built with segments of >= 4 rows
        # (MORIG_CSR_MIN4): the mixed-quad form of the W2-stationary kernel splits a quad that straddles two segments instead of asking
        # for 4-aligned ones -- 14 % (tpl) / 9 % (geo) fewer rows, bit-compatible results, but measured NO faster: under the MFMA load a
        # VALU instruction of the epilogue costs ~15 cycles, and the split quad max + the extra segment pieces eat the rows saved
        # (profiles/r06h_*, r06i_*)
        if (sp and self.WIDTHS[2] == 512 and getattr(csr_tpl, "min4", False) and getattr(csr_geo, "min4", False)
                and os.environ.get("MORIG_EDGE_MIX", "0") == "1"):
            ct, cg = csr_tpl, csr_geo
        self.gcu_3.run(ops, posm, Mat.of(wide, self.X2, self.WIDTHS[1]), ct, cg, Mat.of(wide, self.X3, self.WIDTHS[2]), R, split=sp,
                       pos_feat=pf[2])
        pooled = ops.empty(R * n_graphs, 1024, dev)
        ops.gemm(Mat.of(wide, 0, self.POS), pk["glb"], relu=True, seg=seg, pool=pooled, x_split=sp)
        gb = ops.empty(R * n_graphs, 1024, dev)
        ops.gemm(Mat.of(pooled), pk["g"], relu=False, Y=Mat.of(gb))
        h1 = ops.empty(M, 1024, dev)
        ops.gemm(Mat.of(wide, 0, k_t1), t1, relu=True, Y=Mat.of(h1), rowbias=Mat.of(gb), seg=seg,
                 x_split=sp, y_split=sp)
        h2 = ops.empty(M, 256, dev)
        ops.gemm(Mat.of(h1), pk["t2"], relu=True, Y=Mat.of(h2), x_split=sp, y_split=sp)
        ops.gemm(Mat.of(h2), pk["t3"], relu=False, Y=out, x_split=sp)

    def _forward(self, pos, feature, tpl_edge_index, geo_edge_index, batch):
        ops = get_ops()
        dev = pos.device
        n = pos.shape[0]
        ng = int(batch.max().item()) + 1
        pos4 = torch.zeros((n, 4), dtype=torch.float32, device=dev)
        ops.copy2d(Mat.of(pos.float().contiguous()), Mat.of(pos4, 0, 3))
        feature = feature.unsqueeze(-1) if feature.dim() == 1 else feature
        feat = feature.float().contiguous()
        out = ops.empty(n, self.chn_output, dev)
        self.run(ops, pos4, lambda w, sp: ops.copy2d_pad(Mat.of(feat), w, split=sp), ops.csr_build(tpl_edge_index, n),
                 ops.csr_build(geo_edge_index, n), ops.make_seg(batch, ng, 1), ng, 1, Mat.of(out))
        return out


class _MotionBackbone(NativeModule):
    """Shared front half of JointNetMotion / MaskNetMotion / SkinMotion: motionNet over the keyframes,
    row normalisation, aggregation (models/rignet.py:82-98, 115-131, 194-203)."""

    _RANGE_SCALED = True          # its own pack = the position groups of the GCNRig stacks
    _RANGE_SHIFT_ROOT = True      # its plans scale the stacks' inputs / outputs (range_scale())

    def _pos_units(self):
        """the GCUMotion units whose position branches see data.pos with 3 coordinates and D = 16: motionNet's, then the head's"""
        return [self.motionNet.gcu_1, self.motionNet.gcu_2, self.motionNet.gcu_3]

    def _pack(self):
        import os
        if os.environ.get("MORIG_POS_GROUPS", "1") == "0":
            return dict(pos_groups=(None, [], 0))
        return dict(pos_groups=packing.pack_pos_groups(self._pos_units()))

    def _motion(self, ops, data, input_flow, aggr_method, aggr_out_dim):
        dev = data.pos.device
        n = data.pos.shape[0]
        T = self.num_keyframes
        ng = _num_graphs(data, data.batch)
        flow = input_flow.float().contiguous()
        assert flow.shape[1] >= 3 * T
        pos4 = torch.zeros((n, 4), dtype=torch.float32, device=dev)
        ops.copy2d(Mat.of(data.pos.float().contiguous()), Mat.of(pos4, 0, 3))
        c = range_scale()
        if c != 1.0:
            # range shift [r06]: the GCNRig stacks (and the position branches they share) run at 2^-k -- positions and flows in, every
            # additive constant of their packs (packing.scale_additive), hence every activation; motionNet's output meets F.normalize,
            # which does not see the factor; the head's output is multiplied back (_MotionHead / SkinMotion)
            pos4.mul_(c)
            flow = flow * c
        # both CSR variants of a graph (plain for the narrow layers, 4-aligned for the 128 / 256-wide kernels) from ONE pass over its COO
        csr_tpl, csr_tpl4 = ops.csr_build_dual(data.tpl_edge_index, n)
        csr_geo, csr_geo4 = ops.csr_build_dual(data.geo_edge_index, n)
        seg_T = ops.make_seg(data.batch, ng, T)

        def write_flow(w: Mat, sp: bool):             # feature of replica t = input_flow[:, 3t:3t+3]  (:86), one launch
            ops.copy2d_rep(Mat.of(flow, 0, 3), Mat.of(w.base, w.col0, w.cols, 0, n), T, n, src_col_step=3, split=sp)

        C = self.motionNet.chn_output
        raw = ops.empty(T * n, C, dev)
        # the position branches of motionNet's AND the head's units depend on positions and graphs only: all of them ahead, paired
        # into 32-wide edge layers (packing.pack_pos_groups); a unit without a partner keeps its own 16-wide path
        pos_feats = run_pos_groups(ops, self.packed(dev)["pos_groups"], Mat.of(pos4, 0, 3), csr_tpl, csr_geo)
        pos_feats = pos_feats + [None] * (6 - len(pos_feats))
        # the 3-channel keyframe flows once more as plain fp32 rows [T n, 4]: motionNet's first unit evaluates its first edge Linear from
        # the gathered endpoints (morig_edgeconv_x3) instead of gathering per-vertex [A | B] rows
        # ... as columns 4..7 of rows [pos xyz 0 | flow_t xyz 0]: the same rows are the merged position / feature chunk of mlp_transform's input
        pf8 = ops.empty(T * n, 8, dev)
        ops.copy2d_rep(Mat.of(pos4, 0, 4), Mat.of(pf8, 0, 4, 0, n), T, n)
        ops.copy2d_rep(Mat.of(flow, 0, 3), Mat.of(pf8, 4, 4, 0, n), T, n, src_col_step=3)
        self.motionNet.run(ops, pos4, write_flow, csr_tpl, csr_geo, seg_T, ng, T, Mat.of(raw), csr_geo_wide=csr_geo4,
                           csr_tpl_wide=csr_tpl4, pos_feats=pos_feats[:3], feat3=Mat.of(pf8, 4, 4), posfeat8=Mat.of(pf8))
        motion_all = torch.empty((n, T, C), dtype=torch.float32, device=dev)
        if c != 1.0:
            raw.mul_(1.0 / c)                                         # (F.normalize's eps clamp sees the unscaled rows)
        ops.rownorm(Mat.of(raw), n, T, motion_all, T * C, C)          # F.normalize + torch.stack(dim=1)

        pre = ops.empty(n, aggr_out_dim, dev)
        if aggr_method == "attn":
            self.aggragator.run(ops, motion_all, Mat.of(pre))
        elif aggr_method in ("mean", "max"):
            ops.frame_reduce(motion_all, aggr_method, Mat.of(pre))
        else:
            raise NotImplementedError
        motion_aggr = torch.empty((n, aggr_out_dim), dtype=torch.float32, device=dev)
        ops.rownorm(Mat.of(pre), n, 1, motion_aggr, aggr_out_dim, 0)
        seg_1 = seg_T[:n]
        return dict(pos4=pos4, csr_tpl=csr_tpl, csr_geo=csr_geo, csr_geo4=csr_geo4, csr_tpl4=csr_tpl4, seg=seg_1, ng=ng, motion_all=motion_all,
                    motion_aggr=motion_aggr, pos_feats=pos_feats[3:])


class _MotionHead(_MotionBackbone):
    _head = None

    def _pos_units(self):
        head = getattr(self, self._head)
        return super()._pos_units() + [head.gcu_1, head.gcu_2, head.gcu_3]

    def __init__(self, num_keyframes, chn_output, aggr_method, aggr="max"):
        super().__init__()
        self.num_keyframes = num_keyframes
        self.aggr_method = aggr_method
        self.motionNet = GCNRig(chn_feature=3, chn_output=32, aggr=aggr)
        if self.aggr_method == "attn":
            self.aggragator = TemporalAttn(input_size=32, num_heads=2, hidden_size=64, dim_feedforward=512, output_size=64)
            setattr(self, self._head, GCNRig(chn_feature=64, chn_output=chn_output, aggr=aggr))
        else:
            setattr(self, self._head, GCNRig(chn_feature=32, chn_output=chn_output, aggr=aggr))

    def _forward(self, data, input_flow):
        ops = get_ops()
        head = getattr(self, self._head)
        st = self._motion(ops, data, input_flow, self.aggr_method, head.chn_feature)
        n = data.pos.shape[0]
        aggr = st["motion_aggr"]
        out = torch.empty((n, head.chn_output), dtype=torch.float32, device=aggr.device)
        c = range_scale()
        feat_in = aggr if c == 1.0 else aggr * c                      # (pos4 in `st` is scaled already)
        head.run(ops, st["pos4"], lambda w, sp: ops.copy2d_pad(Mat.of(feat_in), w, split=sp), st["csr_tpl"], st["csr_geo"],
                 st["seg"], st["ng"], 1, Mat.of(out), csr_geo_wide=st["csr_geo4"], csr_tpl_wide=st["csr_tpl4"], pos_feats=st["pos_feats"])
        if c != 1.0:
            out.mul_(1.0 / c)
        return st["motion_all"], aggr, out


    def _forward_train(self, data, input_flow):
        """model.train(): batch-statistics forward, one motionNet pass per keyframe (morig_amd/train_forward.py)"""
        from .. import train_forward
        return train_forward.motion_head_train(get_ops(), self, data, input_flow)

    def _forward_train_grad(self, data, input_flow):
        """model.train() with autograd enabled: the same forward with a graph over the native backward operators"""
        from .. import train_backward
        return train_backward.motion_head_step(self, data, input_flow)


class JointNetMotion(_MotionHead):
    """models/rignet.py:70-100 -> (motion_all, motion_aggr, pred_shift)."""
    _head = "jointnet"


class MaskNetMotion(_MotionHead):
    """models/rignet.py:103-133 -> (motion_all, motion_aggr, pred_mask logits)."""
    _head = "masknet"


class SkinNet_inner(NativeModule):
    """models/rignet.py:136-182."""

    _RANGE_SCALED = True

    def __init__(self, nearest_bone, use_Dg, use_Lf, motion_dim, use_motion, aggr="max"):
        super().__init__()
        self.use_Dg = use_Dg
        self.use_Lf = use_Lf
        self.num_nearest_bone = nearest_bone
        self.input_dim = 3 + nearest_bone * (6 + int(bool(use_Dg)) + int(bool(use_Lf)))
        d = self.input_dim
        self.gcu1 = GCUMotion(in_channels=motion_dim, out_channels=256, in_channel_pos=d, dim_pos_feat=64, aggr=aggr)
        self.gcu2 = GCUMotion(in_channels=256, out_channels=256, in_channel_pos=d, dim_pos_feat=64, aggr=aggr)
        self.gcu3 = GCUMotion(in_channels=256, out_channels=256, in_channel_pos=d, dim_pos_feat=64, aggr=aggr)
        self.multi_layer_tranform2 = MLP([256, 512, 1024])
        self.cls_branch = Sequential(MLP([1024 + 256, 1024, 512]), Linear(512, self.num_nearest_bone))

    def sample_columns(self, total: int):
        """index form of the boolean column selections of :158-171 (8 values per bone:
        6 coords, 1/Dg, leaf flag; drop the switched-off ones, keep the first ``nearest_bone`` bones)."""
        keep = [c for c in range(total) if not ((c % 8 == 6 and not self.use_Dg) or (c % 8 == 7 and not self.use_Lf))]
        return keep[: self.input_dim - 3]

    def _pack(self):
        l1 = self.cls_branch[0][0]
        W = l1[0].weight.detach()                     # input order (:180): [x_3(256) | x_global(1024)]
        c1 = packing.pack_linear(W[:, :256], l1[0].bias, l1[2])
        return dict(
            m1=packing.pack_mlp_layer(self.multi_layer_tranform2[0]),
            m2=packing.pack_mlp_layer(self.multi_layer_tranform2[1]),
            g=packing.couple_rowbias(packing.pack_linear(W[:, 256:]), c1),
            c1=c1,
            c2=packing.pack_mlp_layer(self.cls_branch[0][1]),
            c3=packing.pack_linear(self.cls_branch[1].weight, self.cls_branch[1].bias),
        )

    def run(self, ops, data, motion: torch.Tensor, csr_tpl, csr_geo, seg, n_graphs: int, out: Mat):
        dev = motion.device
        pk = self.packed(dev)
        n = motion.shape[0]
        mdim = motion.shape[1]
        motion = _padded_copy(ops, motion.float())
        P = self.input_dim
        ldp = (P + 3) // 4 * 4
        raw = torch.zeros((n, ldp), dtype=torch.float32, device=dev)          # [pos | selected samples]  (:173)
        ops.copy2d(Mat.of(data.pos.float().contiguous()), Mat.of(raw, 0, 3))
        skin = data.skin_input.float().contiguous()
        cols = torch.tensor(self.sample_columns(skin.shape[1]), dtype=torch.int32, device=dev)
        ops.gather_cols(Mat.of(skin), cols, Mat.of(raw, 3, P - 3))
        c = range_scale()
        if c != 1.0:                                   # range shift: this stack runs at 2^-k too (its caller multiplies `out` back)
            raw.mul_(c)
            motion = motion * c
        posm = Mat.of(raw, 0, P)
        sp = ops.split_activations                    # GEMM -> GEMM activations in the split-fp16 layout
        x1 = ops.empty(n, 256, dev)
        self.gcu1.run(ops, posm, Mat.of(motion, 0, mdim), csr_tpl, csr_geo, Mat.of(x1), split_in=False, split_out=sp)
        g1 = ops.empty(n, 512, dev)
        ops.gemm(Mat.of(x1), pk["m1"], relu=True, Y=Mat.of(g1), x_split=sp, y_split=sp)
        pooled = ops.empty(n_graphs, 1024, dev)
        ops.gemm(Mat.of(g1), pk["m2"], relu=True, seg=seg, pool=pooled, x_split=sp)
        x2 = ops.empty(n, 256, dev)
        self.gcu2.run(ops, posm, Mat.of(x1), csr_tpl, csr_geo, Mat.of(x2), split=sp)
        x3 = ops.empty(n, 256, dev)
        self.gcu3.run(ops, posm, Mat.of(x2), csr_tpl, csr_geo, Mat.of(x3), split=sp)
        gb = ops.empty(n_graphs, 1024, dev)
        ops.gemm(Mat.of(pooled), pk["g"], relu=False, Y=Mat.of(gb))
        h1 = ops.empty(n, 1024, dev)
        ops.gemm(Mat.of(x3), pk["c1"], relu=True, Y=Mat.of(h1), rowbias=Mat.of(gb), seg=seg, x_split=sp, y_split=sp)
        h2 = ops.empty(n, 512, dev)
        ops.gemm(Mat.of(h1), pk["c2"], relu=True, Y=Mat.of(h2), x_split=sp, y_split=sp)
        ops.gemm(Mat.of(h2), pk["c3"], relu=False, Y=out, x_split=sp)

    def _forward(self, data, motion):
        ops = get_ops()
        n = motion.shape[0]
        ng = _num_graphs(data, data.batch)
        motion = motion.float().contiguous()
        out = torch.empty((n, self.num_nearest_bone), dtype=torch.float32, device=motion.device)
        self.run(ops, data, motion, ops.csr_build(data.tpl_edge_index, n), ops.csr_build(data.geo_edge_index, n),
                 ops.make_seg(data.batch, ng, 1), ng, Mat.of(out))
        return out


class SkinMotion(_MotionBackbone):
    """models/rignet.py:185-205 -> (motion_all, motion_aggr, skin_cls_pred)."""

    def __init__(self, nearest_bone, use_Dg, use_Lf, num_keyframes, use_motion, motion_dim, aggr="max"):
        super().__init__()
        self.num_keyframes = num_keyframes
        self.motion_dim = motion_dim
        self.motionNet = GCNRig(chn_feature=3, chn_output=motion_dim, aggr=aggr)
        self.aggragator = TemporalAttn(input_size=motion_dim, num_heads=2, hidden_size=64, dim_feedforward=512,
                                       output_size=motion_dim)
        self.skinNet = SkinNet_inner(nearest_bone, use_Dg, use_Lf, motion_dim, use_motion, aggr)

    def _forward(self, data, input_flow):
        ops = get_ops()
        st = self._motion(ops, data, input_flow, "attn", self.motion_dim)
        n = data.pos.shape[0]
        aggr = st["motion_aggr"]
        out = torch.empty((n, self.skinNet.num_nearest_bone), dtype=torch.float32, device=aggr.device)
        self.skinNet.run(ops, data, aggr, st["csr_tpl4"], st["csr_geo4"], st["seg"], st["ng"], Mat.of(out))     # all three GCUs are 256 wide
        if range_scale() != 1.0:
            out.mul_(1.0 / range_scale())
        return st["motion_all"], aggr, out

    def _forward_train(self, data, input_flow):
        """model.train(): batch-statistics forward (morig_amd/train_forward.py)"""
        from .. import train_forward
        return train_forward.skin_motion_train(get_ops(), self, data, input_flow)

    def _forward_train_grad(self, data, input_flow):
        from .. import train_backward
        return train_backward.skin_motion_step(self, data, input_flow)


def jointnet_motion(**kwargs):
    return JointNetMotion(num_keyframes=kwargs["num_keyframes"], chn_output=kwargs["chn_output"],
                          aggr_method=kwargs["aggr_method"])


def masknet_motion(**kwargs):
    return MaskNetMotion(num_keyframes=kwargs["num_keyframes"], chn_output=kwargs["chn_output"],
                         aggr_method=kwargs["aggr_method"])


def skinnet_motion(**kwargs):
    return SkinMotion(nearest_bone=kwargs["nearest_bone"], use_Dg=kwargs["use_Dg"], use_Lf=kwargs["use_Lf"],
                      num_keyframes=kwargs["num_keyframes"], use_motion=kwargs["use_motion"],
                      motion_dim=kwargs["motion_dim"])
