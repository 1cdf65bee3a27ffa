"""Train-mode FORWARD of the rignet family on the MI355X-native op layer (forward half).

In ``model.train()`` every ``BatchNorm1d`` of the reference normalises with the statistics of the current batch and updates
its running buffers (momentum 0.1, unbiased variance, ``num_batches_tracked += 1``): over the vertices of the batch in the
dense MLPs, over the EDGES in the per-edge MLPs of EdgeConvMotion (/root/reference/models/basic_modules.py:31-36, 179-202;
/root/reference/training/train_rig.py:136-195). The eval path's fusions (BatchNorm folded into the next Linear at pack time,
keyframes batched as replicas, max taken inside the EdgeConv kernel) all assume FIXED affines, so the train-mode forward runs
layer by layer instead:

    dense layer   morig_gemm (Linear + ReLU on MFMA) -> morig_col_stats -> running-stat update -> morig_col_affine
    edge MLP      per-vertex GEMM ([A | B] = [(W_a - W_b) x + b | W_b x]) -> morig_edge_gather_relu (+ the BN1 statistics over
                  edges, same pass) -> morig_edge_hidden (hidden affine + Linear2 + ReLU on MFMA) -> morig_col_stats (BN2 over edges)
                  -> morig_segmax_affine (max over incoming edges behind the BatchNorm affine)
    keyframes     one motionNet pass per keyframe, as the reference does: each pass has its own batch statistics and moves the
                  running buffers once (models/rignet.py:85-88)

What this is NOT: the backward pass. Outputs carry no autograd graph; ``training/train_rig.py`` still cannot optimise through
it. CorrNet / DeformNet have no train-mode path yet.
"""
from __future__ import annotations

import torch
from torch.nn import BatchNorm1d

from . import packing
from .native import CSR, Mat


# ---- cross-rank batch statistics (SURVEY 8(e) caveat / f-4): a train-mode forward is only shardable over GPUs if every
# BatchNorm sees the statistics of the GLOBAL batch. With a process group set here, the per-rank moments are combined by ONE
# all-reduce of (sum x, sum x^2, n) per BatchNorm (float64, 2C + 1 numbers; RCCL on GPUs, gloo in the CPU tests), and the
# backward combines its two sums (sum dz, sum dz xhat) the same way: exactly torch.nn.SyncBatchNorm's scheme.
_SYNC = {"group": None}


def set_batchnorm_sync(group) -> None:
    """group: a torch.distributed process group, True for the default group, None to switch the synchronisation off"""
    _SYNC["group"] = group


def _all_reduce(buf: torch.Tensor) -> torch.Tensor:
    import torch.distributed as dist
    g = _SYNC["group"]
    dist.all_reduce(buf, group=None if g is True else g)
    return buf


def batch_moments(ops, X: Mat, rows_dev=None, local=None):
    """-> (mean, biased var, count [1], local share n_local / n) of the rows of X over the whole (cross-rank) batch; ``local``: this
    rank's (mean, var, count) when the pass that wrote X already took them (``edge_gather_relu(want_stats=True)``)"""
    mean, var, cnt = local if local is not None else ops.col_stats(X, rows_dev=rows_dev)
    if _SYNC["group"] is None:
        return mean, var, cnt, None
    C = mean.numel()
    n_loc = cnt.double()
    buf = _all_reduce(torch.cat([mean.double() * n_loc, (var.double() + mean.double() ** 2) * n_loc, n_loc]))
    n = buf[-1].clamp(min=1.0)
    m = buf[:C] / n
    v = (buf[C:2 * C] / n - m * m).clamp(min=0.0)
    return m.float(), v.float(), n.float().reshape(1), (n_loc / n).float().reshape(1)


def sync_backward_sums(sdz: torch.Tensor, sdzx: torch.Tensor, share):
    """the backward's two BatchNorm sums over the global batch, pre-scaled by this rank's share of the rows: the kernels divide by
    the LOCAL row count, and (global sum) * (n_local / n) / n_local = (global sum) / n. -> (sums for the kernel, global sums)"""
    if _SYNC["group"] is None or share is None:
        return sdz, sdzx, sdz, sdzx
    C = sdz.numel()
    buf = _all_reduce(torch.cat([sdz.double(), sdzx.double()]))
    gs, gx = buf[:C].float(), buf[C:].float()
    return (gs * share).contiguous(), (gx * share).contiguous(), gs, gx


def _bn_train(bn: BatchNorm1d, mean: torch.Tensor, var: torch.Tensor, count: torch.Tensor, want_rstd: bool = False):
    """batch-statistics affine (s, t) of a BatchNorm1d in training mode + its running-buffer update
    (torch.nn.functional.batch_norm semantics: biased variance normalises, unbiased variance is tracked); want_rstd: also
    1 / sqrt(var + eps) for the backward. One native launch (morig_bn_finalize) when the layer has a numeric momentum --
    the reference's MLP() has 0.1; the cumulative-average form (momentum=None) reads the batch counter on the host and stays torch."""
    tracking = bn.track_running_stats and bn.running_mean is not None
    if bn.momentum is not None or not tracking:
        from .runtime import get_ops
        s, t, rstd = get_ops().bn_finalize(bn, mean.contiguous(), var.contiguous(), count)
        return (s, t, rstd) if want_rstd else (s, t)
    with torch.no_grad():
        s = bn.weight.detach().float() / torch.sqrt(var + bn.eps)
        t = bn.bias.detach().float() - mean * s
        bn.num_batches_tracked += 1
        m = 1.0 / float(bn.num_batches_tracked)
        unbiased = var * (count / torch.clamp(count - 1.0, min=1.0))
        bn.running_mean.mul_(1.0 - m).add_(mean, alpha=m)
        bn.running_var.mul_(1.0 - m).add_(unbiased, alpha=m)
    return (s.contiguous(), t.contiguous(), torch.rsqrt(var + bn.eps)) if want_rstd else (s.contiguous(), t.contiguous())


def _pad_to(v: torch.Tensor, n: int, fill: float) -> torch.Tensor:
    out = torch.full((n,), fill, dtype=torch.float32, device=v.device)
    out[: v.numel()] = v
    return out


def _ld4(c: int) -> int:
    return (c + 3) // 4 * 4


def dense_train(ops, X: Mat, layer, out: Mat = None) -> torch.Tensor:
    """one ``Seq(Linear, ReLU, BatchNorm1d)`` of MLP() on the rows of X with batch statistics; returns / fills [rows, N]."""
    lin, bn = layer[0], layer[2]
    dev = X.base.device
    pk = packing.to_device(packing.pack_linear(lin.weight, lin.bias), dev)
    N = pk.N
    if out is None:
        buf = ops.empty(X.rows, _ld4(N), dev)
        out = Mat.of(buf, 0, N)
    ops.gemm(X, pk, relu=True, Y=out)
    mean, var, cnt, _ = batch_moments(ops, out)
    s, t = _bn_train(bn, mean, var, cnt)
    ops.col_affine(out, s, t)
    return out.base


def edge_mlp_train(ops, X: Mat, csr: CSR, mlp, out: Mat) -> None:
    """per-edge ``MLP([2C, H, H])`` on [x_i ‖ x_j - x_i] with BatchNorm statistics over the edges, max over incoming edges
    (models/basic_modules.py:153-155 / 192-195 in training mode). ``csr`` must be the UNPADDED CSR (every edge counted once)."""
    assert not csr.quad, "batch statistics over edges need every edge exactly once"
    dev = X.base.device
    l1, l2 = mlp[0], mlp[1]
    W1 = l1[0].weight.detach().float()
    H, C = W1.shape[0], W1.shape[1] // 2
    vertex = packing.to_device(packing.pack_linear(torch.cat([W1[:, :C] - W1[:, C:], W1[:, C:]], 0),
                                                   torch.cat([l1[0].bias.detach().float(), torch.zeros(H, device=W1.device)], 0)), dev)
    n = X.rows
    ab = ops.empty(n, _ld4(2 * H), dev)
    ops.gemm(X, vertex, relu=False, Y=Mat.of(ab, 0, 2 * H))
    A, B = Mat.of(ab, 0, H), Mat.of(ab, H, H)
    e_live = csr.rowptr[csr.n_nodes:csr.n_nodes + 1]                       # E' on the device
    z1 = ops.empty(csr.capacity, _ld4(H), dev)
    loc1 = ops.edge_gather_relu(A, B, csr, Mat.of(z1, 0, H), want_stats=True)
    mean1, var1, cnt, _ = batch_moments(ops, Mat.of(z1, 0, H), rows_dev=e_live, local=loc1)
    s1, t1 = _bn_train(l1[2], mean1, var1, cnt)
    Hp, Kp = max(H, 32), (H + 31) // 32 * 32
    W2 = torch.zeros((Hp, Kp), dtype=torch.float32, device=dev)
    W2[:H, :H] = l2[0].weight.detach().float()
    ones, zeros = torch.ones(Hp, device=dev), torch.zeros(Hp, device=dev)
    pe = packing.PackedEdge(H, _pad_to(s1, Kp, 1.0), _pad_to(t1, Kp, 0.0), W2.contiguous(), _pad_to(l2[0].bias.detach().float(), Hp, 0.0),
                            ones, zeros, None)                              # hidden affine applied while gathering: fp32 MFMA path
    z2 = ops.empty(csr.capacity, _ld4(H), dev)
    ops.edge_hidden(A, B, csr, pe, Mat.of(z2, 0, H))
    mean2, var2, cnt2, _ = batch_moments(ops, Mat.of(z2, 0, H), rows_dev=e_live)
    s2, t2 = _bn_train(l2[2], mean2, var2, cnt2)
    ops.segmax_affine(Mat.of(z2, 0, H), csr.rowptr, csr.n_nodes, out, s2, t2)


def edgeconvmotion_train(ops, ec, pos: Mat, x: Mat, csr: CSR, out: Mat) -> None:
    """EdgeConvMotion (models/basic_modules.py:179-202): out = [ nn_x branch (O/2) | nn_pos branch (D) ]"""
    H = ec.nn_x[1][0].weight.shape[0]
    D = ec.nn_pos[1][0].weight.shape[0]
    edge_mlp_train(ops, x, csr, ec.nn_x, Mat.of(out.base, out.col0, H, out.row0, out.rows))
    edge_mlp_train(ops, pos, csr, ec.nn_pos, Mat.of(out.base, out.col0 + H, D, out.row0, out.rows))


def gcumotion_train(ops, gcu, pos: Mat, x: Mat, csr_tpl: CSR, csr_geo: CSR, out: Mat) -> None:
    """GCUMotion (models/basic_modules.py:205-219): two EdgeConvMotions, concatenated, then MLP([O + 2D, O])."""
    dev = x.base.device
    n = x.rows
    H = gcu.edge_conv_tpl.nn_x[1][0].weight.shape[0]
    D = gcu.edge_conv_tpl.nn_pos[1][0].weight.shape[0]
    cat = ops.empty(n, _ld4(2 * (H + D)), dev)
    edgeconvmotion_train(ops, gcu.edge_conv_tpl, pos, x, csr_tpl, Mat.of(cat, 0, H + D))
    edgeconvmotion_train(ops, gcu.edge_conv_geo, pos, x, csr_geo, Mat.of(cat, H + D, H + D))
    dense_train(ops, Mat.of(cat, 0, 2 * (H + D)), gcu.mlp[0], out)


def gcnrig_train(ops, net, pos4: torch.Tensor, feature: torch.Tensor, csr_tpl: CSR, csr_geo: CSR, batch_i32: torch.Tensor,
                 mesh_ptr: torch.Tensor, n_graphs: int) -> torch.Tensor:
    """GCNRig.forward (models/rignet.py:59-67) with batch statistics. pos4: [n, 4] (xyz, 0); feature: [n, F]; -> [n, chn_output]."""
    dev = pos4.device
    n, F = pos4.shape[0], feature.shape[1]
    w1, w2, w3 = net.WIDTHS
    tr = getattr(net, net.TRANSFORM)
    feat = feature.float().contiguous()
    featp = feat if feat.shape[1] % 4 == 0 else torch.nn.functional.pad(feat, (0, _ld4(F) - F))    # 16-byte aligned GEMM operand
    posm = Mat.of(pos4, 0, 3)
    xcat = ops.empty(n, w1 + w2 + w3, dev)                                 # [x_1 | x_2 | x_3]  (:62)
    gcumotion_train(ops, net.gcu_1, posm, Mat.of(featp, 0, F), csr_tpl, csr_geo, Mat.of(xcat, 0, w1))
    gcumotion_train(ops, net.gcu_2, posm, Mat.of(xcat, 0, w1), csr_tpl, csr_geo, Mat.of(xcat, w1, w2))
    gcumotion_train(ops, net.gcu_3, posm, Mat.of(xcat, w1, w2), csr_tpl, csr_geo, Mat.of(xcat, w1 + w2, w3))
    x4 = dense_train(ops, Mat.of(xcat), net.mlp_glb[0])
    xg = ops.empty(n_graphs, 1024, dev)
    ops.segmax_affine(Mat.of(x4, 0, 1024), mesh_ptr, n_graphs, Mat.of(xg))          # scatter_max over each mesh (:63)
    # x_5 (:65): [x_global(1024) | pos(3) | feature(F) | x_1 | x_2 | x_3]
    k5 = 1024 + 3 + F + w1 + w2 + w3
    x5 = torch.zeros((n, _ld4(k5)), dtype=torch.float32, device=dev)
    ops.gather_rows(Mat.of(xg), batch_i32, Mat.of(x5, 0, 1024))                      # repeat_interleave (:64)
    ops.copy2d(posm, Mat.of(x5, 1024, 3))
    ops.copy2d(Mat.of(featp, 0, F), Mat.of(x5, 1027, F))
    ops.copy2d(Mat.of(xcat), Mat.of(x5, 1027 + F, w1 + w2 + w3))
    h1 = dense_train(ops, Mat.of(x5, 0, k5), tr[0][0])
    h2 = dense_train(ops, Mat.of(h1, 0, tr[0][0][0].out_features), tr[0][1])
    last = packing.to_device(packing.pack_linear(tr[1].weight, tr[1].bias), dev)
    out = ops.empty(n, _ld4(last.N), dev)
    ops.gemm(Mat.of(h2, 0, tr[0][1][0].out_features), last, relu=False, Y=Mat.of(out, 0, last.N))
    return out[:, :last.N]


def temporalattn_train(ops, attn, x: torch.Tensor) -> torch.Tensor:
    """TemporalAttn.forward (models/rignet.py:36-46): the attention has no BatchNorm (same kernels as eval); the feed-forward
    MLP([hidden, dim_feedforward, output]) runs with batch statistics over the vertices."""
    dev = x.device
    pk = attn.packed(dev)
    n = x.shape[0]
    y = ops.empty(n, pk["g"].shape[0] * x.shape[2], dev)
    ops.cls_attention(x, pk["g"], pk["cls"], Mat.of(y))
    res = ops.empty(n, pk["mix"].N, dev)
    ops.gemm(Mat.of(y), pk["mix"], relu=False, Y=Mat.of(res))
    h = dense_train(ops, Mat.of(res), attn.feedforward[0])
    return dense_train(ops, Mat.of(h, 0, attn.feedforward[0][0].out_features), attn.feedforward[1])


def _graph_state(ops, data):
    dev = data.pos.device
    n = data.pos.shape[0]
    batch = data.batch
    ng = getattr(data, "num_graphs", None)
    if ng is None:
        ng = int(batch.max().item()) + 1
    pos4 = torch.zeros((n, 4), dtype=torch.float32, device=dev)
    ops.copy2d(Mat.of(data.pos.float().contiguous()), Mat.of(pos4, 0, 3))
    counts = torch.bincount(batch, minlength=ng)
    mesh_ptr = torch.zeros(ng + 1, dtype=torch.int32, device=dev)
    mesh_ptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
    return dict(pos4=pos4, n=n, ng=int(ng), csr_tpl=ops.csr_build(data.tpl_edge_index, n), csr_geo=ops.csr_build(data.geo_edge_index, n),
                batch_i32=batch.to(torch.int32).contiguous(), mesh_ptr=mesh_ptr)


def motion_backbone_train(ops, model, data, input_flow, aggr_method, out_dim):
    """keyframe loop + normalisation + aggregation (models/rignet.py:82-98): one motionNet pass PER keyframe -- each has its own
    batch statistics and moves the BatchNorm running buffers once, exactly as the reference's Python loop does."""
    st = _graph_state(ops, data)
    dev, n = data.pos.device, st["n"]
    T = model.num_keyframes
    flow = input_flow.float().contiguous()
    C = model.motionNet.chn_output
    motion_all = torch.empty((n, T, C), dtype=torch.float32, device=dev)
    for t in range(T):
        raw = gcnrig_train(ops, model.motionNet, st["pos4"], flow[:, 3 * t:3 * t + 3], st["csr_tpl"], st["csr_geo"], st["batch_i32"],
                           st["mesh_ptr"], st["ng"]).contiguous()
        ops.rownorm(Mat.of(raw), n, 1, motion_all.view(-1)[t * C:], T * C, 0)      # F.normalize + its slot of torch.stack (:87-88)
    if aggr_method == "attn":
        pre = temporalattn_train(ops, model.aggragator, motion_all)[:, :out_dim].contiguous()
    elif aggr_method in ("mean", "max"):
        pre = ops.empty(n, out_dim, dev)
        ops.frame_reduce(motion_all, aggr_method, Mat.of(pre))
    else:
        raise NotImplementedError
    motion_aggr = torch.empty((n, out_dim), dtype=torch.float32, device=dev)
    ops.rownorm(Mat.of(pre), n, 1, motion_aggr, out_dim, 0)
    st.update(motion_all=motion_all, motion_aggr=motion_aggr)
    return st


def motion_head_train(ops, model, data, input_flow):
    """JointNetMotion / MaskNetMotion in training mode -> (motion_all, motion_aggr, head output)."""
    head = getattr(model, model._head)
    st = motion_backbone_train(ops, model, data, input_flow, model.aggr_method, head.chn_feature)
    out = gcnrig_train(ops, head, st["pos4"], st["motion_aggr"], st["csr_tpl"], st["csr_geo"], st["batch_i32"], st["mesh_ptr"], st["ng"])
    return st["motion_all"], st["motion_aggr"], out.contiguous()


def skinnet_train(ops, net, data, motion: torch.Tensor, st) -> torch.Tensor:
    """SkinNet_inner.forward (models/rignet.py:158-182) with batch statistics."""
    dev = motion.device
    n = motion.shape[0]
    P = net.input_dim
    raw = torch.zeros((n, _ld4(P)), dtype=torch.float32, device=dev)      # raw_input = [pos | selected skin samples]  (:173)
    ops.copy2d(Mat.of(data.pos.float().contiguous()), Mat.of(raw, 0, 3))
    skin = data.skin_input.float().contiguous()
    cols = torch.tensor(net.sample_columns(skin.shape[1]), dtype=torch.int32, device=dev)
    ops.gather_cols(Mat.of(skin), cols, Mat.of(raw, 3, P - 3))
    posm = Mat.of(raw, 0, P)
    mdim = motion.shape[1]
    mo = motion.float().contiguous()
    mo = mo if mdim % 4 == 0 else torch.nn.functional.pad(mo, (0, _ld4(mdim) - mdim))
    x1 = ops.empty(n, 256, dev)
    gcumotion_train(ops, net.gcu1, posm, Mat.of(mo, 0, mdim), st["csr_tpl"], st["csr_geo"], Mat.of(x1))
    g1 = dense_train(ops, Mat.of(x1), net.multi_layer_tranform2[0])
    g2 = dense_train(ops, Mat.of(g1, 0, 512), net.multi_layer_tranform2[1])
    xg = ops.empty(st["ng"], 1024, dev)
    ops.segmax_affine(Mat.of(g2, 0, 1024), st["mesh_ptr"], st["ng"], Mat.of(xg))
    x2 = ops.empty(n, 256, dev)
    gcumotion_train(ops, net.gcu2, posm, Mat.of(x1), st["csr_tpl"], st["csr_geo"], Mat.of(x2))
    x4 = ops.empty(n, 256 + 1024, dev)                                     # [x_3 | x_global]  (:180)
    gcumotion_train(ops, net.gcu3, posm, Mat.of(x2), st["csr_tpl"], st["csr_geo"], Mat.of(x4, 0, 256))
    ops.gather_rows(Mat.of(xg), st["batch_i32"], Mat.of(x4, 256, 1024))
    cb = net.cls_branch
    h1 = dense_train(ops, Mat.of(x4), cb[0][0])
    h2 = dense_train(ops, Mat.of(h1, 0, 1024), cb[0][1])
    last = packing.to_device(packing.pack_linear(cb[1].weight, cb[1].bias), dev)
    out = ops.empty(n, _ld4(last.N), dev)
    ops.gemm(Mat.of(h2, 0, 512), last, relu=False, Y=Mat.of(out, 0, last.N))
    return out[:, :last.N].contiguous()


def skin_motion_train(ops, model, data, input_flow):
    """SkinMotion in training mode -> (motion_all, motion_aggr, skin_cls_pred)."""
    st = motion_backbone_train(ops, model, data, input_flow, "attn", model.motion_dim)
    return st["motion_all"], st["motion_aggr"], skinnet_train(ops, model.skinNet, data, st["motion_aggr"], st)
