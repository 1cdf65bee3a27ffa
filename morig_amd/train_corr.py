"""Train-mode forward AND backward of CorrNet and DeformNet on the MI355X-native op layer (the
callers training/train_corr_pose.py:61-70, train_corr_shape.py, train_deform_pose.py:29-40, train_deform_shape.py).

Composition of the autograd blocks of ``train_backward.py`` -- ``DenseTrain`` (one ``Seq(Linear, ReLU, BatchNorm1d)`` with batch
statistics: native forward, native dX / dW / BatchNorm backward), ``EdgeMLPTrain`` (EdgeConv's per-edge MLP with statistics over
the edges and arg-max routed max aggregation), ``SegMaxPool`` / ``RowGather`` (pooling and its broadcast), ``NativeLinear`` (``linear``) -- in
the order of /root/reference/models/corrnet.py:37-77 and models/deformnet.py:40-99:

  vertex branch   GCU x 4 (models/basic_modules.py:165-177: two EdgeConvs on [x_i ‖ x_j - x_i], concatenated, MLP) -> mlp_glb ->
                  per-mesh max -> broadcast -> vtx_mlp -> F.normalize
  point branch    SAModule x 3 (:66-86): FPS (native, indices only) -> ball query (native slot table, the deterministic
                  ``radius`` of the GPU branch) -> PointConv = local_nn on the rows [x_j ‖ pos_j - pos_i] of the bipartite graph
                  (PyG's self-loop quirk included: the CSR is the eval path's) with BatchNorm statistics over those rows, max per
                  centre; GlobalSAModule (:115-125); FPModule x 4 (:127-138): knn_interpolate weights and indices from the native
                  search (positions carry no gradient), the interpolation itself is three gathered rows per target in torch
  matching        cosine 1-NN indices from the native kernel on the detached features (an arg-max has no gradient), the
                  combined row [v ‖ p ‖ <v, p>] differentiable as in the reference's GPU branch (:64-65), then lin_vismask

BatchNorm running buffers move once per layer, as in the reference. Everything that is a contraction or a per-row statistic runs
in the native operators; index bookkeeping, concatenations, F.normalize and the interpolation sums stay torch autograd on device
tensors. There is no CPU fallback: the op layer raises without the HIP library.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .native import Mat
from .runtime import get_ops
from .train_backward import RowGather, SegMaxPool, _canonical_csr_of, _guarded_step, edge_mlp, gcnrig, linear, mlp_layer


def _mlp(x, layers):
    for layer in layers:
        x = mlp_layer(x, layer)
    return x


def _pos4(pos: torch.Tensor) -> torch.Tensor:
    p4 = torch.zeros((pos.shape[0], 4), dtype=torch.float32, device=pos.device)
    p4[:, :3] = pos.detach().float()
    return p4


def gcu(unit, x, csr_tpl, csr_geo):
    """GCU.forward (models/basic_modules.py:172-177); the EdgeConv MLP attribute is called ``nn_pos`` (:145) but acts on features"""
    both = torch.cat([edge_mlp(x, csr_tpl, unit.edge_conv_tpl.nn_pos), edge_mlp(x, csr_geo, unit.edge_conv_geo.nn_pos)], 1)
    return mlp_layer(both, unit.mlp[0])


def vertex_branch(net, data, st):
    vtx = data.vtx.float()
    x1 = gcu(net.vtx_gcu_1, vtx, st["csr_tpl"], st["csr_geo"])
    x2 = gcu(net.vtx_gcu_2, x1, st["csr_tpl"], st["csr_geo"])
    x3 = gcu(net.vtx_gcu_3, x2, st["csr_tpl"], st["csr_geo"])
    x4 = gcu(net.vtx_gcu_4, x3, st["csr_tpl"], st["csr_geo"])
    cat = torch.cat([x1, x2, x3, x4], 1)
    g = SegMaxPool.apply(mlp_layer(cat, net.vtx_mlp_glb[0]), st["ptr_v"], st["B"])
    x6 = torch.cat([RowGather.apply(g, st["vtx_batch"], st["B"]), vtx, cat], 1)                    # corrnet.py:45-46
    h = _mlp(x6, net.vtx_mlp[0])
    return F.normalize(linear(h, net.vtx_mlp[1]), dim=1)


def set_abstraction(mod, x, pos4, pos_new4, ptr, out_ptr, B):
    """SAModule's convolution (models/basic_modules.py:76-84) on the given sampling: -> x_new [M, H3]"""
    ops = get_ops()
    N, M = pos4.shape[0], pos_new4.shape[0]
    coo = ops.ball_query(Mat.of(pos4, 0, 3), ptr, Mat.of(pos_new4, 0, 3), out_ptr, B, mod.r, mod.max_num_neighbors)
    csr = ops.csr_from_slots(coo, M, mod.max_num_neighbors, N)
    E = int(csr.rowptr[M].item())                                            # live rows of the bipartite graph (host read: sizes a tensor)
    src, dst = csr.src[:E].long(), csr.dst[:E].long()
    rel = pos4[src, :3] - pos_new4[dst, :3]
    rows = rel if x is None else torch.cat([x.index_select(0, src), rel], 1)  # PointConv.message: local_nn([x_j ‖ pos_j - pos_i])
    return SegMaxPool.apply(_mlp(rows, mod.conv.local_nn), csr.rowptr, M)


def interpolate(feat, nn):
    """knn_interpolate (models/basic_modules.py:134): sum_s w_s x[idx_s] / sum_s w_s, w = 1 / clamp(d^2, 1e-16) from the native
    search; slots without a source (clouds smaller than k) carry index -1"""
    idx, wgt = nn
    w = torch.where(idx >= 0, wgt, torch.zeros_like(wgt)).unsqueeze(-1)     # [ny, 3, 1]
    g = feat.index_select(0, idx.clamp(min=0).reshape(-1).long()).reshape(idx.shape[0], idx.shape[1], -1)
    return (g * w).sum(1) / w.sum(1)


def point_branch(net, data, plan):
    ops = get_ops()
    B = plan.B
    pos0 = _pos4(data.pts)
    levels = [pos0]
    for level in range(3):                                                   # fps -> pos[idx]  (:75, :85)
        src = levels[-1]
        M = sum(plan.counts[level + 1])
        idx = ops.fps(Mat.of(src, 0, 3), plan.ptr[level], plan.ptr[level + 1], plan.start[level], B, max(plan.counts[level]), M)
        nxt = torch.zeros((M, 4), dtype=torch.float32, device=src.device)
        nxt[:, :3] = src[idx.long(), :3]
        levels.append(nxt)
    pos0, pos1, pos2, pos3 = levels
    ptr0, ptr1, ptr2, ptr3 = plan.ptr
    x1 = set_abstraction(net.pts_sa1_module, None, pos0, pos1, ptr0, ptr1, B)
    x2 = set_abstraction(net.pts_sa2_module, x1, pos1, pos2, ptr1, ptr2, B)
    x3 = set_abstraction(net.pts_sa3_module, x2, pos2, pos3, ptr2, ptr3, B)
    pooled = SegMaxPool.apply(_mlp(torch.cat([x3, pos3[:, :3]], 1), net.pts_sa4_module.nn), ptr3, B)   # GlobalSAModule (:121-122)
    batch3 = torch.repeat_interleave(torch.arange(B, device=pos0.device), (ptr3[1:] - ptr3[:-1]).long(), output_size=pos3.shape[0])
    # FP4 interpolates with k = 1 from the ONE pooled point of the cloud: the broadcast row ((x w) / w: equal to <= 1 ulp)
    f4 = _mlp(torch.cat([RowGather.apply(pooled, batch3, B), x3], 1), net.pts_fp4_module.nn)

    def propagate(fp, feat, lvl, skip):
        nn = fp.search(ops, levels[lvl], plan.ptr[lvl], levels[lvl - 1], plan.ptr[lvl - 1], B, max(plan.counts[lvl - 1]))
        y = interpolate(feat, nn)
        return _mlp(y if skip is None else torch.cat([y, skip], 1), fp.nn)

    f3 = propagate(net.pts_fp3_module, f4, 3, x2)
    f2 = propagate(net.pts_fp2_module, f3, 2, x1)
    f1 = propagate(net.pts_fp1_module, f2, 1, None)
    h = _mlp(f1, net.pts_mlp[0])
    return F.normalize(linear(h, net.pts_mlp[1]), dim=1)


def _state(net, data, random_start):
    """offsets, FPS starts and the two unpadded CSRs of one batch (the graph build's status words are read by the caller)"""
    from .models.corrnet import _HostPlan
    ops = get_ops()
    dev = data.vtx.device
    B = getattr(data, "num_graphs", None)
    vb, pb = data.vtx_batch, data.pts_batch
    if B is None:
        B = int(max(int(vb.max().item()), int(pb.max().item()))) + 1
    vcounts, pcounts = torch.stack([torch.bincount(vb, minlength=B), torch.bincount(pb, minlength=B)]).tolist()
    plan = _HostPlan(vcounts, pcounts, [m.ratio for m in (net.pts_sa1_module, net.pts_sa2_module, net.pts_sa3_module)], random_start, dev)
    n = data.vtx.shape[0]
    return dict(B=B, n=n, plan=plan, ptr_v=plan.ptr_v, vtx_batch=vb.long(), vcounts=vcounts,
                csr_tpl=_canonical_csr_of(data.tpl_edge_index, n), csr_geo=_canonical_csr_of(data.geo_edge_index, n))


def _guarded(data, net, random_start, body):
    """train_backward._guarded_step with this file's batch state: CSR status words read BEFORE any BatchNorm buffer moves, range
    flag checked after"""
    return _guarded_step(data, body, state=lambda d: _state(net, d, random_start), dev=data.vtx.device)


def _corrnet(net, data, train_vismask, st):
    ops = get_ops()
    out_vtx = vertex_branch(net, data, st)
    out_pts = point_branch(net, data, st["plan"])
    vis = None
    if train_vismask:
        nn, _ = ops.cosine_nn(Mat.of(out_vtx.detach().contiguous()), st["ptr_v"], Mat.of(out_pts.detach().contiguous()), st["plan"].ptr[0],
                              st["B"], max(st["vcounts"]))
        b = out_pts.index_select(0, nn.long())
        comb = torch.cat([out_vtx, b, (out_vtx * b).sum(1, keepdim=True)], 1)                    # corrnet.py:65
        h = _mlp(comb, net.lin_vismask[0])
        vis = linear(h, net.lin_vismask[1])
    return out_vtx, out_pts, vis


def corrnet_step(net, data, train_vismask, random_start=True):
    """CorrNet.forward in training mode with an autograd graph (models/corrnet.py:37-77) -> (out_vtx, out_pts, out_vismask, temprature)"""
    def body(st):
        return _corrnet(net, data, train_vismask, st) + (net.temprature,)
    return _guarded(data, net, random_start, body)


def deformnet_step(model, data):
    """DeformNet.forward in training mode with an autograd graph (models/deformnet.py:40-99)
    -> (pred_flow, vtx_feature, pts_feature, pred_vismask, tau)"""
    ops = get_ops()
    net = model.corr_extractor
    k = int(model.num_interp)

    def body(st):
        vtx_f, pts_f, logit = _corrnet(net, data, True, st)
        B, n = st["B"], st["n"]
        vb = st["vtx_batch"]
        ptr_v, ptr_p = st["ptr_v"], st["plan"].ptr[0]
        vis = torch.sigmoid(logit)                                                                  # :41-46, per mesh min-max
        lo = -SegMaxPool.apply(-vis, ptr_v, B)
        hi = SegMaxPool.apply(vis, ptr_v, B)
        vis = (vis - lo.index_select(0, vb)) / (hi - lo).index_select(0, vb)
        # visible part (:49-54): every vertex votes from its k most similar points; the indices carry no gradient
        vf, pf = vtx_f.detach().contiguous(), pts_f.detach().contiguous()
        idx = ops.cosine_knn(Mat.of(vf), ptr_v, Mat.of(pf), ptr_p, B, max(st["vcounts"]), k).long()          # [n, k], -1 padded
        live = (idx >= 0).unsqueeze(-1).float()
        j = idx.clamp(min=0)
        dist = data.pts.float()[j] - data.vtx.float().unsqueeze(1)                                # [n, k, 3]
        sim = (pts_f[j] * vtx_f.unsqueeze(1)).sum(-1, keepdim=True) * vis.unsqueeze(1) * live     # :52-53
        # (the vertex whose normalised mask is exactly 0 has sum(sim) = 0: the reference's 0 / 0 row, overwritten by the invisible
        # vote below; taken out of the quotient so that no NaN enters the backward pass through the rows that ARE kept)
        den = sim.sum(1)
        ok = den != 0
        flow = torch.where(ok, (dist * sim).sum(1) / torch.where(ok, den, torch.ones_like(den)), torch.full_like(den, float("nan")))
        # invisible part (:57-95): vertices with mask < 0.5 vote from their k most similar visible vertices of the mesh
        vmask = vis.detach().contiguous()
        idx2 = ops.cosine_knn(Mat.of(vf), ptr_v, Mat.of(vf), ptr_v, B, max(st["vcounts"]), k, vis=Mat.of(vmask), split=True).long()
        invis = (vmask < 0.5).squeeze(1)
        if bool(invis.any()):
            j2 = idx2[invis]
            live2 = (j2 >= 0).unsqueeze(-1).float()
            j2 = j2.clamp(min=0)
            sim2 = (vtx_f[j2] * vtx_f[invis].unsqueeze(1)).sum(-1, keepdim=True) * live2
            fj = torch.where(live2.bool(), flow[j2], torch.zeros((), device=flow.device))       # (a padded slot points at row 0, weight 0)
            inv_flow = (fj * sim2).sum(1) / sim2.sum(1)                                           # votes come from VISIBLE rows of `flow`
            flow = flow.index_put((invis.nonzero(as_tuple=True)[0],), inv_flow)
        model.last_neighbours = (idx.int(), idx2.int())           # [n, k] tables (-1 padded), as the eval forward leaves them
        l1 = torch.cat([flow, vis], 1)
        pred = gcnrig(model.completing, data.vtx.float(), l1, st["csr_tpl"], st["csr_geo"], vb, ptr_v, B)
        return pred, vtx_f, pts_f, vis, net.temprature
    return _guarded(data, net, True, body)
