"""TEST INFRASTRUCTURE ONLY -- never imported by the product (``morig_amd``).

CPU oracle: a from-scratch, plain PyTorch fp32 restatement of the MoRig geometric-network
forward path, straight-line and unoptimised on purpose (per-edge MLPs exactly as the reference
evaluates them -- none of the product's algebraic restructurings), with ``state_dict`` keys
identical to the reference so the same parameters drive both.

Each class cites the reference lines it restates. It is checked against the reference's own
models/*.py (imported through oracle/shim.py in the build container) by tests/golden/*.npz; see
oracle/make_golden.py. Allowed importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import pyg_primitives as P


def mlp_stack(widths: Sequence[int]) -> nn.Sequential:
    """Reference ``MLP(channels)``: per layer Linear -> ReLU -> BatchNorm1d, BN *after* ReLU
    (models/basic_modules.py:31-36). Keys: ``{layer}.0.*`` Linear, ``{layer}.2.*`` BN."""
    layers = []
    for cin, cout in zip(widths[:-1], widths[1:]):
        layers.append(nn.Sequential(nn.Linear(cin, cout), nn.ReLU(), nn.BatchNorm1d(cout, momentum=0.1)))
    return nn.Sequential(*layers)


def _loop_normalised(edge_index: torch.Tensor, n: int) -> torch.Tensor:
    """strip existing self loops, append exactly one per node (basic_modules.py:149-150,188-189)."""
    ei, _ = P.remove_self_loops(edge_index)
    ei, _ = P.add_self_loops(ei, num_nodes=n)
    return ei


class EdgeMaxConv(nn.Module):
    """``EdgeConv`` (models/basic_modules.py:142-163): message nn_pos([x_i ‖ x_j - x_i]) for every
    edge j->i (j = edge_index[0], i = edge_index[1]); per-channel max at i."""

    def __init__(self, cin: int, cout: int):
        super().__init__()
        self.nn_pos = mlp_stack([2 * cin, cout, cout])

    def forward(self, x, edge_index):
        ei = _loop_normalised(edge_index, x.size(0))
        xj, xi = x[ei[0]], x[ei[1]]
        msg = self.nn_pos(torch.cat([xi, xj - xi], dim=1))
        return P.propagate_max(msg, ei[1], x.size(0))


class GraphConvUnit(nn.Module):
    """``GCU`` (models/basic_modules.py:165-177)."""

    def __init__(self, cin: int, cout: int):
        super().__init__()
        self.edge_conv_tpl = EdgeMaxConv(cin, cout // 2)
        self.edge_conv_geo = EdgeMaxConv(cin, cout // 2)
        self.mlp = mlp_stack([cout, cout])

    def forward(self, x, tpl, geo):
        return self.mlp(torch.cat([self.edge_conv_tpl(x, tpl), self.edge_conv_geo(x, geo)], dim=1))


class EdgeMaxConvMotion(nn.Module):
    """``EdgeConvMotion`` (models/basic_modules.py:179-202): message
    nn_x([x_i ‖ x_j - x_i]) ‖ nn_pos([pos_i ‖ pos_j - pos_i]); max at i; 1-D x is unsqueezed."""

    def __init__(self, cin: int, chalf: int, cpos: int, dpos: int):
        super().__init__()
        self.nn_x = mlp_stack([2 * cin, chalf, chalf])
        self.nn_pos = mlp_stack([2 * cpos, dpos, dpos])

    def forward(self, pos, x, edge_index):
        if x.dim() == 1:
            x = x.unsqueeze(-1)
        ei = _loop_normalised(edge_index, x.size(0))
        j, i = ei[0], ei[1]
        fx = self.nn_x(torch.cat([x[i], x[j] - x[i]], dim=1))
        fp = self.nn_pos(torch.cat([pos[i], pos[j] - pos[i]], dim=1))
        return P.propagate_max(torch.cat([fx, fp], dim=1), i, x.size(0))


class GraphConvUnitMotion(nn.Module):
    """``GCUMotion`` (models/basic_modules.py:205-219)."""

    def __init__(self, cin: int, cout: int, cpos: int = 3, dpos: int = 16):
        super().__init__()
        self.edge_conv_tpl = EdgeMaxConvMotion(cin, cout // 2, cpos, dpos)
        self.edge_conv_geo = EdgeMaxConvMotion(cin, cout // 2, cpos, dpos)
        self.mlp = mlp_stack([cout + 2 * dpos, cout])

    def forward(self, pos, x, tpl, geo):
        both = torch.cat([self.edge_conv_tpl(pos, x, tpl), self.edge_conv_geo(pos, x, geo)], dim=1)
        return self.mlp(both)


def _pool_and_broadcast(x: torch.Tensor, batch: torch.Tensor) -> torch.Tensor:
    """scatter_max over meshes, then repeat_interleave back to vertices (rignet.py:63-64)."""
    g, _ = P.scatter_max(x, batch, dim=0)
    return torch.repeat_interleave(g, torch.bincount(batch), dim=0)


class RigGCN(nn.Module):
    """``GCNRig`` (models/rignet.py:49-67)."""

    def __init__(self, chn_feature: int, chn_output: int):
        super().__init__()
        self.gcu_1 = GraphConvUnitMotion(chn_feature, 64)
        self.gcu_2 = GraphConvUnitMotion(64, 256)
        self.gcu_3 = GraphConvUnitMotion(256, 512)
        self.mlp_glb = mlp_stack([64 + 256 + 512, 1024])
        self.mlp_transform = nn.Sequential(mlp_stack([1024 + 3 + chn_feature + 832, 1024, 256]),
                                           nn.Linear(256, chn_output))

    def forward(self, pos, feature, tpl, geo, batch):
        a = self.gcu_1(pos, feature, tpl, geo)
        b = self.gcu_2(pos, a, tpl, geo)
        c = self.gcu_3(pos, b, tpl, geo)
        g = _pool_and_broadcast(self.mlp_glb(torch.cat([a, b, c], dim=1)), batch)
        return self.mlp_transform(torch.cat([g, pos, feature, a, b, c], dim=1))


class ClsTemporalAttention(nn.Module):
    """``TemporalAttn`` (models/rignet.py:10-46): CLS token prepended, multi-head scaled dot-product
    attention (hidden size is *per head*), w_o, only token 0 kept, then MLP([hid, ff, out])."""

    def __init__(self, input_size, num_heads, hidden_size, dim_feedforward, output_size):
        super().__init__()
        self.num_heads = num_heads
        self.w_qs = nn.Linear(input_size, hidden_size * num_heads, bias=False)
        self.w_ks = nn.Linear(input_size, hidden_size * num_heads, bias=False)
        self.w_vs = nn.Linear(input_size, hidden_size * num_heads, bias=False)
        self.w_o = nn.Linear(hidden_size * num_heads, hidden_size, bias=False)
        self.feedforward = mlp_stack([hidden_size, dim_feedforward, output_size])
        self.cls_token = nn.Parameter(torch.randn(1, 1, input_size))

    def forward(self, x):                      # x: V x T x C
        V = x.shape[0]
        tok = torch.cat([self.cls_token.expand(V, -1, -1), x], dim=1)      # V x (T+1) x C
        nh = self.num_heads

        def heads(t):                          # V x L x (nh*d) -> (V*nh) x L x d
            L = t.shape[1]
            return t.reshape(V, L, nh, -1).permute(0, 2, 1, 3).reshape(V * nh, L, -1)

        q, k, v = heads(self.w_qs(tok)), heads(self.w_ks(tok)), heads(self.w_vs(tok))
        att = torch.softmax(torch.bmm(q, k.transpose(1, 2)) / math.sqrt(k.size(-1)), dim=-1)
        res = torch.bmm(att, v)                # (V*nh) x L x d
        L = res.shape[1]
        res = res.reshape(V, nh, L, -1).permute(0, 2, 1, 3).reshape(V, L, -1)
        return self.feedforward(self.w_o(res)[:, 0, :])


class _MotionHeadNet(nn.Module):
    """Shared body of ``JointNetMotion`` / ``MaskNetMotion`` (models/rignet.py:70-133); the two
    differ only in the attribute name of the head GCNRig."""
    head_name = "head"

    def __init__(self, num_keyframes, chn_output, aggr_method, aggr="max"):
        super().__init__()
        self.num_keyframes = num_keyframes
        self.aggr_method = aggr_method
        self.motionNet = RigGCN(3, 32)
        if aggr_method == "attn":
            self.aggragator = ClsTemporalAttention(32, 2, 64, 512, 64)
            setattr(self, self.head_name, RigGCN(64, chn_output))
        else:
            setattr(self, self.head_name, RigGCN(32, chn_output))

    def forward(self, data, input_flow):
        tpl, geo, batch = data.tpl_edge_index, data.geo_edge_index, data.batch
        frames = []
        for t in range(self.num_keyframes):
            m = self.motionNet(data.pos, input_flow[:, 3 * t:3 * t + 3], tpl, geo, batch)
            frames.append(F.normalize(m, dim=1))
        motion_all = torch.stack(frames, dim=1)
        if self.aggr_method == "attn":
            aggr = self.aggragator(motion_all)
        elif self.aggr_method == "mean":
            aggr = motion_all.mean(dim=1)
        elif self.aggr_method == "max":
            aggr = motion_all.max(dim=1)[0]
        else:
            raise NotImplementedError
        aggr = F.normalize(aggr, dim=1)
        out = getattr(self, self.head_name)(data.pos, aggr, tpl, geo, batch)
        return motion_all, aggr, out


class JointNetMotion(_MotionHeadNet):
    head_name = "jointnet"


class MaskNetMotion(_MotionHeadNet):
    head_name = "masknet"


def skin_input_columns(nearest_bone: int, use_Dg: bool, use_Lf: bool, total: int = 160) -> List[int]:
    """Column selection of ``data.skin_input`` (models/rignet.py:158-171) as an index list:
    per bone 8 values [6 coords, 1/Dg, leaf]; drop Dg (col 6) and/or Lf (col 7) per flags,
    then keep the first ``nearest_bone`` bones."""
    cols = []
    for c in range(total):
        r = c % 8
        if r == 6 and not use_Dg:
            continue
        if r == 7 and not use_Lf:
            continue
        cols.append(c)
    per_bone = 6 + int(use_Dg) + int(use_Lf)
    return cols[: per_bone * nearest_bone]


class SkinNet_inner(nn.Module):
    """models/rignet.py:136-182."""

    def __init__(self, nearest_bone, use_Dg, use_Lf, motion_dim, use_motion, aggr="max"):
        super().__init__()
        self.use_Dg, self.use_Lf, self.num_nearest_bone = use_Dg, use_Lf, nearest_bone
        cpos = 3 + nearest_bone * (6 + int(use_Dg) + int(use_Lf))
        self.gcu1 = GraphConvUnitMotion(motion_dim, 256, cpos=cpos, dpos=64)
        self.gcu2 = GraphConvUnitMotion(256, 256, cpos=cpos, dpos=64)
        self.gcu3 = GraphConvUnitMotion(256, 256, cpos=cpos, dpos=64)
        self.multi_layer_tranform2 = mlp_stack([256, 512, 1024])
        self.cls_branch = nn.Sequential(mlp_stack([1024 + 256, 1024, 512]), nn.Linear(512, nearest_bone))

    def forward(self, data, motion):
        cols = skin_input_columns(self.num_nearest_bone, self.use_Dg, self.use_Lf, data.skin_input.shape[1])
        raw = torch.cat([data.pos, data.skin_input[:, cols]], dim=1)
        tpl, geo = data.tpl_edge_index, data.geo_edge_index
        x1 = self.gcu1(raw, motion, tpl, geo)
        g = _pool_and_broadcast(self.multi_layer_tranform2(x1), data.batch)
        x2 = self.gcu2(raw, x1, tpl, geo)
        x3 = self.gcu3(raw, x2, tpl, geo)
        return self.cls_branch(torch.cat([x3, g], dim=1))


class SkinMotion(nn.Module):
    """models/rignet.py:185-205."""

    def __init__(self, nearest_bone, use_Dg, use_Lf, num_keyframes, use_motion, motion_dim, aggr="max"):
        super().__init__()
        self.num_keyframes, self.motion_dim = num_keyframes, motion_dim
        self.motionNet = RigGCN(3, motion_dim)
        self.aggragator = ClsTemporalAttention(motion_dim, 2, 64, 512, motion_dim)
        self.skinNet = SkinNet_inner(nearest_bone, use_Dg, use_Lf, motion_dim, use_motion)

    def forward(self, data, input_flow):
        tpl, geo, batch = data.tpl_edge_index, data.geo_edge_index, data.batch
        frames = [F.normalize(self.motionNet(data.pos, input_flow[:, 3 * t:3 * t + 3], tpl, geo, batch), dim=1)
                  for t in range(self.num_keyframes)]
        motion_all = torch.stack(frames, dim=1)
        aggr = F.normalize(self.aggragator(motion_all), dim=1)
        return motion_all, aggr, self.skinNet(data, aggr)


# ----------------------------------------------------------------------------- CorrNet
class SetAbstraction(nn.Module):
    """``SAModule`` (models/basic_modules.py:66-86), GPU-branch semantics (deterministic
    ``radius``: first 64 hits in index order, strict <) -- the branch the reference runs on a GPU."""

    def __init__(self, ratio, r, net, max_num_neighbors):
        super().__init__()
        self.ratio, self.r, self.max_num_neighbors = ratio, r, max_num_neighbors
        self.conv = P.PointConv(net)

    def forward(self, x, pos, batch, random_start=True):
        idx = P.fps(pos, batch, ratio=self.ratio, random_start=random_start)
        row, col = P.radius(pos, pos[idx], self.r, batch, batch[idx], max_num_neighbors=self.max_num_neighbors)
        ei = torch.stack([col, row], dim=0)
        xs = (None, None) if x is None else (x, x[idx])
        return self.conv(xs, (pos, pos[idx]), ei), pos[idx], batch[idx]


class GlobalSetAbstraction(nn.Module):
    """``GlobalSAModule`` (models/basic_modules.py:115-125)."""

    def __init__(self, net):
        super().__init__()
        self.nn = net

    def forward(self, x, pos, batch):
        g = P.global_max_pool(self.nn(torch.cat([x, pos], dim=1)), batch)
        return g, pos.new_zeros((g.size(0), 3)), torch.arange(g.size(0), device=batch.device)


class FeaturePropagation(nn.Module):
    """``FPModule`` (models/basic_modules.py:127-138)."""

    def __init__(self, k, net):
        super().__init__()
        self.k = k
        self.nn = net

    def forward(self, x, pos, batch, x_skip, pos_skip, batch_skip):
        y = P.knn_interpolate(x, pos, pos_skip, batch, batch_skip, k=self.k)
        if x_skip is not None:
            y = torch.cat([y, x_skip], dim=1)
        return self.nn(y), pos_skip, batch_skip


class CorrNet(nn.Module):
    """models/corrnet.py:10-77 (GPU-branch semantics for radius / cosine 1-NN)."""

    def __init__(self, input_feature, output_feature, temprature, aggr="max"):
        super().__init__()
        self.input_feature, self.output_feature = input_feature, output_feature
        self.temprature = nn.Parameter(torch.tensor([float(temprature)]))
        self.vtx_gcu_1 = GraphConvUnit(3, 32)
        self.vtx_gcu_2 = GraphConvUnit(32, 64)
        self.vtx_gcu_3 = GraphConvUnit(64, 256)
        self.vtx_gcu_4 = GraphConvUnit(256, 512)
        self.vtx_mlp_glb = mlp_stack([864, 1024])
        self.vtx_mlp = nn.Sequential(mlp_stack([1024 + 3 + 864, 1024, 256]), nn.Linear(256, output_feature))
        self.pts_sa1_module = SetAbstraction(0.5, 0.12, mlp_stack([input_feature, 32, 32, 64]), 64)
        self.pts_sa2_module = SetAbstraction(0.25, 0.25, mlp_stack([64 + 3, 64, 64, 128]), 64)
        self.pts_sa3_module = SetAbstraction(0.25, 0.5, mlp_stack([128 + 3, 256, 256, 256]), 64)
        self.pts_sa4_module = GlobalSetAbstraction(mlp_stack([256 + 3, 256, 256, 512]))
        self.pts_fp4_module = FeaturePropagation(1, mlp_stack([512 + 256, 256, 256]))
        self.pts_fp3_module = FeaturePropagation(3, mlp_stack([256 + 128, 256, 128]))
        self.pts_fp2_module = FeaturePropagation(3, mlp_stack([128 + 64, 128, 64]))
        self.pts_fp1_module = FeaturePropagation(3, mlp_stack([64, 64, 64]))
        self.pts_mlp = nn.Sequential(mlp_stack([64, 64]), nn.Linear(64, output_feature))
        self.lin_vismask = nn.Sequential(mlp_stack([2 * output_feature + 1, 256, 128, 64]), nn.Linear(64, 1))

    def vertex_branch(self, data):
        tpl, geo = data.tpl_edge_index, data.geo_edge_index
        x1 = self.vtx_gcu_1(data.vtx, tpl, geo)
        x2 = self.vtx_gcu_2(x1, tpl, geo)
        x3 = self.vtx_gcu_3(x2, tpl, geo)
        x4 = self.vtx_gcu_4(x3, tpl, geo)
        cat = torch.cat([x1, x2, x3, x4], dim=1)
        g = _pool_and_broadcast(self.vtx_mlp_glb(cat), data.vtx_batch)
        return F.normalize(self.vtx_mlp(torch.cat([g, data.vtx, cat], dim=1)), dim=1)

    def point_branch(self, data, random_start):
        s0 = (None, data.pts, data.pts_batch)
        s1 = self.pts_sa1_module(*s0, random_start)
        s2 = self.pts_sa2_module(*s1, random_start)
        s3 = self.pts_sa3_module(*s2, random_start)
        s4 = self.pts_sa4_module(*s3)
        f4 = self.pts_fp4_module(*s4, *s3)
        f3 = self.pts_fp3_module(*f4, *s2)
        f2 = self.pts_fp2_module(*f3, *s1)
        f1, _, _ = self.pts_fp1_module(*f2, *s0)
        return F.normalize(self.pts_mlp(f1), dim=1)

    def forward(self, data, train_vismask, random_start=True):
        out_vtx = self.vertex_branch(data)
        out_pts = self.point_branch(data, random_start)
        vis = None
        if train_vismask:
            yi, xi = P.knn(out_pts, out_vtx, 1, data.pts_batch, data.vtx_batch, cosine=True)
            a, b = out_vtx[yi], out_pts[xi]
            vis = self.lin_vismask(torch.cat([a, b, (a * b).sum(dim=1, keepdim=True)], dim=1))
        return out_vtx, out_pts, vis, self.temprature


class DeformGCN(nn.Module):
    """``GCNDeform`` (models/deformnet.py:13-32): GCNRig's wiring at widths 128/256/512; note the
    argument order (geo before tpl) and the attribute name ``mlp_tramsform`` (sic)."""

    def __init__(self, chn_in: int, chn_output: int):
        super().__init__()
        self.gcu_1 = GraphConvUnitMotion(chn_in, 128)
        self.gcu_2 = GraphConvUnitMotion(128, 256)
        self.gcu_3 = GraphConvUnitMotion(256, 512)
        self.mlp_glb = mlp_stack([128 + 256 + 512, 1024])
        self.mlp_tramsform = nn.Sequential(mlp_stack([1024 + 3 + chn_in + 896, 1024, 256]), nn.Linear(256, chn_output))

    def forward(self, pos, feature, geo, tpl, batch):
        a = self.gcu_1(pos, feature, tpl, geo)
        b = self.gcu_2(pos, a, tpl, geo)
        c = self.gcu_3(pos, b, tpl, geo)
        g = _pool_and_broadcast(self.mlp_glb(torch.cat([a, b, c], dim=1)), batch)
        return self.mlp_tramsform(torch.cat([g, pos, feature, a, b, c], dim=1))


def _vote(values: torch.Tensor, weights: torch.Tensor, target: torch.Tensor, n: int) -> torch.Tensor:
    """scatter_add(values * w) / scatter_add(w) over the neighbours of every target (deformnet.py:54,94);
    a target without neighbours (or whose weights sum to 0) gets 0/0 = NaN, as in the reference."""
    return P.scatter_add(values * weights, target, dim=0, dim_size=n) / P.scatter_add(weights, target, dim=0, dim_size=n)


class DeformNet(nn.Module):
    """models/deformnet.py:35-99: CorrNet features -> sigmoid visibility, min-max normalised per mesh (:42-46) ->
    flow of visible vertices voted from their ``num_interp`` most similar points (:49-54) -> flow of invisible
    vertices voted from their most similar *visible* vertices (:57-95) -> GCNDeform on [flow_init | vismask] (:97-98)."""

    def __init__(self, tau_nce, num_interp):
        super().__init__()
        self.corr_extractor = CorrNet(3, 64, temprature=tau_nce)
        self.completing = DeformGCN(chn_in=4, chn_output=3)
        self.num_interp = num_interp

    @staticmethod
    def _pairs(lists: torch.Tensor):
        """[n, k] neighbour table (-1 padded) -> (query, neighbour) index vectors in knn's order."""
        q, t = torch.nonzero(lists >= 0, as_tuple=True)
        return q, lists[q, t].long()

    def forward(self, data, neighbours=None):
        """neighbours (tests only): (points of every vertex [n, k], visible vertices of every vertex [n, k]; global
        indices, -1 padded) to use instead of the two knn calls -- similarity near-ties make the k-NN choice
        ill-conditioned on symmetric meshes, so large-size parity is checked downstream of a fixed choice."""
        vtx_f, pts_f, vis, tau = self.corr_extractor(data, True)          # random_start defaults to True (:41)
        vis = torch.sigmoid(vis)
        for s, e in P._segments(data.vtx_batch):
            m = vis[s:e]
            vis[s:e] = (m - m.min()) / (m.max() - m.min())
        n, k = vtx_f.shape[0], self.num_interp
        if neighbours is not None:
            return self._with_neighbours(data, vtx_f, pts_f, vis, tau, *neighbours)
        yi, xi = P.knn(pts_f, vtx_f, k, data.pts_batch, data.vtx_batch, cosine=True)
        sim = (pts_f[xi] * vtx_f[yi]).sum(dim=-1, keepdim=True) * vis[yi]
        flow = _vote(data.pts[xi] - data.vtx[yi], sim, yi, n)
        seen = (vis >= 0.5).squeeze(1)
        hidden = (vis < 0.5).squeeze(1)                                   # NaN masks fall in neither set
        seen_ids, hidden_ids = torch.nonzero(seen).squeeze(1), torch.nonzero(hidden).squeeze(1)
        if hidden_ids.numel():
            yi2, xi2 = P.knn(vtx_f[seen], vtx_f[hidden], k, data.vtx_batch[seen], data.vtx_batch[hidden], cosine=True)
            sim2 = (vtx_f[seen][xi2] * vtx_f[hidden][yi2]).sum(dim=-1, keepdim=True)
            flow[hidden_ids] = _vote(flow[seen_ids][xi2], sim2, yi2, hidden_ids.numel())
        pred = self.completing(data.vtx, torch.cat([flow, vis], dim=-1), data.geo_edge_index, data.tpl_edge_index,
                               data.vtx_batch)
        return pred, vtx_f, pts_f, vis, tau

    def _with_neighbours(self, data, vtx_f, pts_f, vis, tau, to_points, to_visible):
        n = vtx_f.shape[0]
        yi, xi = self._pairs(to_points)
        sim = (pts_f[xi] * vtx_f[yi]).sum(dim=-1, keepdim=True) * vis[yi]
        flow = _vote(data.pts[xi] - data.vtx[yi], sim, yi, n)
        hidden_ids = torch.nonzero((vis < 0.5).squeeze(1)).squeeze(1)
        yi2, xi2 = self._pairs(to_visible)
        assert bool((vis[yi2] < 0.5).all()) and bool((vis[xi2] >= 0.5).all())
        sim2 = (vtx_f[xi2] * vtx_f[yi2]).sum(dim=-1, keepdim=True)
        voted = _vote(flow[xi2], sim2, yi2, n)
        flow[hidden_ids] = voted[hidden_ids]
        pred = self.completing(data.vtx, torch.cat([flow, vis], dim=-1), data.geo_edge_index, data.tpl_edge_index,
                               data.vtx_batch)
        return pred, vtx_f, pts_f, vis, tau


# ----------------------------------------------------------------------------- factories
def jointnet_motion(**kw):
    return JointNetMotion(kw["num_keyframes"], kw["chn_output"], kw["aggr_method"])


def masknet_motion(**kw):
    return MaskNetMotion(kw["num_keyframes"], kw["chn_output"], kw["aggr_method"])


def skinnet_motion(**kw):
    return SkinMotion(kw["nearest_bone"], kw["use_Dg"], kw["use_Lf"], kw["num_keyframes"],
                      kw["use_motion"], kw["motion_dim"])


def corrnet(**kw):
    return CorrNet(kw["input_feature"], kw["output_feature"], kw["temprature"])


def deformnet(**kw):
    return DeformNet(kw["tau_nce"], kw["num_interp"])
