"""TEST DOUBLE (tests/ only): a torch-CPU emulation of the op interface in morig_amd/native.py.

It exists so the HOST logic of the product -- parameter packing, column placement in the wide
activation buffers, replica handling, plan wiring -- can be checked against the oracle on a machine
without a GPU. It mimics what each HIP kernel computes from the *packed* structures (including the
zero-padding conventions), not what the reference computes. The product never imports this file;
without the HIP library the product raises.
"""
from __future__ import annotations

import torch

from morig_amd.native import CSR, Mat


class EmuOps:
    name = "emulated"
    precision = "f32"

    def guarded(self, device, fn, rerun=True, retry=None):
        return fn()

    @property
    def split_activations(self):
        return self.emulate_split

    def edgeconv_pair(self, first, second):
        self.edgeconv(**first)
        self.edgeconv(**second)

    def edgeconv_x3_pair(self, first, second):
        self.edgeconv_x3(**first)
        self.edgeconv_x3(**second)

    def empty(self, rows, cols, device, dtype=torch.float32):
        # poison, so a plan that reads something it never wrote fails loudly
        return torch.full((rows, cols), float("nan"), dtype=dtype, device=device)

    # -- graph ----------------------------------------------------------------------------------
    def csr_build(self, edge_index, n_nodes, n_src=None, skip_negative=False, pad4=False, min4=False):
        src, dst = edge_index[0].long(), edge_index[1].long()
        if skip_negative:
            keep = (src >= 0) & (dst >= 0)
            src, dst = src[keep], dst[keep]
        n_src = n_nodes if n_src is None else n_src
        assert int(src.min()) >= 0 and int(src.max()) < n_src and int(dst.min()) >= 0 and int(dst.max()) < n_nodes
        keep = src != dst
        loop = torch.arange(n_nodes)
        src = torch.cat([src[keep], loop])
        dst = torch.cat([dst[keep], loop])
        if pad4:        # pad every target's segment to a multiple of 4 with copies of its self loop
            deg = torch.bincount(dst, minlength=n_nodes)
            extra = torch.repeat_interleave(loop, (-deg) % 4)
            src, dst = torch.cat([src, extra]), torch.cat([dst, extra])
        if min4:        # MORIG_CSR_MIN4: fill every segment up to 4 rows with copies of its self loop
            deg = torch.bincount(dst, minlength=n_nodes)
            extra = torch.repeat_interleave(loop, (4 - deg).clamp(min=0))
            src, dst = torch.cat([src, extra]), torch.cat([dst, extra])
        order = torch.sort(dst, stable=True)[1]
        src, dst = src[order], dst[order]
        rowptr = torch.zeros(n_nodes + 1, dtype=torch.int64)
        rowptr[1:] = torch.cumsum(torch.bincount(dst, minlength=n_nodes), 0)
        cap = edge_index.shape[1] + (4 if (pad4 or min4) else 1) * n_nodes
        pad = cap - src.numel()
        junk = torch.full((pad,), -12345, dtype=torch.int32)
        return CSR(rowptr.int(), torch.cat([src.int(), junk]), torch.cat([dst.int(), junk]), n_nodes, cap,
                   torch.zeros(1, dtype=torch.int32), edge_count=int(src.numel()), quad=pad4, min4=min4)

    def csr_build_dual(self, edge_index, n_nodes, min4=False):
        return self.csr_build(edge_index, n_nodes, min4=min4), self.csr_build(edge_index, n_nodes, pad4=True)

    def csr_from_slots(self, coo, n_nodes, max_nbrs, n_src):
        assert coo.shape == (2, n_nodes * max_nbrs)
        return self.csr_build(coo, n_nodes, n_src=n_src, skip_negative=True)

    # -- dense -------------------------------------------------------------------------------------
    @staticmethod
    def _x(X: Mat, K: int):
        assert X.ld % 4 == 0 and X.col0 % 4 == 0, "GEMM operand rows must be 16-byte aligned"
        assert X.cols == K
        v = X.view()
        assert not torch.isnan(v).any(), "GEMM reads uninitialised memory"
        return v

    fast = False
    emulate_split = False       # tests set this to exercise the split-activation plan wiring (alignment rules only)

    def copy2d_pad(self, src: Mat, dst: Mat, split=False):
        if split:
            assert dst.col0 % 32 == 0 and dst.cols % 32 == 0 and dst.ld % 32 == 0, "split slot alignment"
        d = dst.view()
        d.zero_()
        d[:, : src.cols] = src.view()

    def copy2d_rep(self, src: Mat, dst: Mat, replicas, dst_row_step, src_col_step=0, split=False):
        for r in range(replicas):
            self.copy2d_pad(Mat.of(src.base, src.col0 + r * src_col_step, src.cols, src.row0, src.rows),
                            Mat.of(dst.base, dst.col0, dst.cols, dst.row0 + r * dst_row_step, dst.rows), split=split)

    def gemm_takes_tail(self, lin, Y: Mat, n_tail_cols):
        # the library's rule (csrc/gemm_dma.hip): K tails on the 256 x 256 LDS-DMA store kernel only
        return bool(self.emulate_split and lin.N % 256 == 0 and lin.K % 32 == 0 and n_tail_cols % 32 == 0 and lin.K > n_tail_cols
                    and Y.col0 % 4 == 0 and Y.ld % 4 == 0)

    def pack_tails(self, src, col_a, col_b, wa, wb):
        assert wa + wb <= 32 and len(col_a) == len(col_b) <= 8
        out = torch.zeros((len(col_a), src.shape[0], 32), dtype=torch.float32)
        for t, (a, b) in enumerate(zip(col_a, col_b)):
            out[t, :, :wa] = src[:, a:a + wa]
            out[t, :, wa:wa + wb] = src[:, b:b + wb]
        assert not torch.isnan(out).any(), "pack_tails reads uninitialised memory"
        return out

    def gemm(self, X: Mat, lin, relu, Y=None, rowbias=None, seg=None, pool=None, affine=True, x_split=False, y_split=False, x_tail=None):
        if x_tail is not None:
            # morig_gemm_args.X_tail: the last x_tail.cols input columns from row (row % x_tail.rows) of the tail matrix
            assert x_split and pool is None and self.gemm_takes_tail(lin, Y, x_tail.cols), "x_tail where the library would refuse it"
            assert x_tail.col0 % 32 == 0 and x_tail.ld % 32 == 0 and X.cols + x_tail.cols == lin.K
            full = torch.cat([X.view(), x_tail.view()[torch.arange(X.rows) % x_tail.rows]], dim=1).contiguous()
            assert X.col0 % 32 == 0 and X.ld % 32 == 0 and not torch.isnan(full).any()
            return self.gemm(Mat.of(full), lin, relu, Y=Y, rowbias=rowbias, seg=seg, pool=pool, affine=affine, y_split=y_split)
        if x_split:
            assert X.col0 % 32 == 0 and X.ld % 32 == 0, "split-fp16 X window must be chunk aligned"
            # logical K may end inside a chunk: the remaining columns of that chunk must hold finite data
            tail = X.base[X.row0:X.row0 + X.rows, X.col0 + X.cols: X.col0 + (X.cols + 31) // 32 * 32]
            assert not torch.isnan(tail).any(), "split-fp16 X: chunk tail uninitialised"
        if y_split:
            assert Y.col0 % 32 == 0 and Y.ld % 32 == 0 and pool is None
        x = self._x(X, lin.K)
        acc = x @ lin.W[: lin.N, : lin.K].t()
        assert float(lin.W[lin.N:].abs().sum()) == 0 and float(lin.W[:, lin.K:].abs().sum()) == 0, "padding must be zero"
        acc = acc + lin.bias[: lin.N]
        if rowbias is not None:
            acc = acc + rowbias.view()[seg.long()][:, : lin.N]
        if relu:
            acc = torch.relu(acc)
        if affine and lin.scale is not None:
            acc = acc * lin.scale[: lin.N] + lin.shift[: lin.N]
        if pool is not None:
            assert Y is None
            out = torch.full_like(pool, float("nan"))
            idx = seg.long()
            for s in torch.unique(idx):
                out[s, : lin.N] = acc[idx == s].max(dim=0)[0]
            pool.copy_(out)
        else:
            Y.view().copy_(acc)

    # -- fused edge conv ---------------------------------------------------------------------------
    def edgeconv_can_split_out(self, A: Mat, B: Mat, csr: CSR, ec, out: Mat, replicas=1, in_rep_stride=0, out_rep_stride=0):
        # the library's rule (tile_gemm.hip edge_plan): the 4-aligned-CSR kernels at H = 128 / 256, chunk-aligned output window
        return bool(self.emulate_split and (csr.quad or (getattr(csr, "min4", False) and ec.H == 256)) and ec.H in (128, 256) and ec.s1 is None
                    and out.col0 % 32 == 0 and out.ld % 32 == 0)

    def edgeconv(self, A: Mat, B: Mat, csr: CSR, ec, out: Mat, replicas=1, in_rep_stride=0, out_rep_stride=0, out_split=False):
        assert A.ld % 4 == 0 and A.col0 % 4 == 0 and B.ld % 4 == 0 and B.col0 % 4 == 0
        if out_split:
            assert self.edgeconv_can_split_out(A, B, csr, ec, out), "out_split where the library would refuse it"
        H = ec.H
        E = int(csr.rowptr[-1])
        src, dst = csr.src[:E].long(), csr.dst[:E].long()
        n = csr.n_nodes
        for r in range(replicas):
            a = A.base[A.row0 + r * in_rep_stride + dst, A.col0:A.col0 + H]
            b = B.base[B.row0 + r * in_rep_stride + src, B.col0:B.col0 + H]
            assert not torch.isnan(a).any() and not torch.isnan(b).any()
            h1 = torch.relu(a + b)
            if ec.s1 is not None:
                h1 = h1 * ec.s1[:H] + ec.t1[:H]
            z = torch.relu(h1 @ ec.W2[:H, :H].t() + ec.b2[:H]) * ec.s2[:H] + ec.t2[:H]
            res = torch.full((n, H), float("-inf"))
            res = res.scatter_reduce(0, dst[:, None].expand(-1, H), z, reduce="amax", include_self=True)
            out.base[out.row0 + r * out_rep_stride: out.row0 + r * out_rep_stride + n, out.col0:out.col0 + H] = res

    def edgeconv_x3(self, X: Mat, first, csr: CSR, ec, out: Mat, replicas=1, in_rep_stride=0, out_rep_stride=0):
        """first Linear on the gathered endpoints' 3 channels, then the 32-wide layer: A = W1a x + b1, B = W1b x per vertex"""
        W1a, W1b, b1 = first
        rows = X.base.shape[0] - X.row0
        x = X.base[X.row0:X.row0 + rows, X.col0:X.col0 + 3]
        ab = torch.zeros(rows, 64)
        ab[:, :32] = x @ W1a[:, :3].t() + b1
        ab[:, 32:] = x @ W1b[:, :3].t()
        self.edgeconv(Mat.of(ab, 0, 32), Mat.of(ab, 32, 32), csr, ec, out, replicas, in_rep_stride, out_rep_stride)

    # -- small ops -----------------------------------------------------------------------------------
    def copy2d(self, src: Mat, dst: Mat):
        dst.view().copy_(src.view())

    def gather_cols(self, src: Mat, cols, dst: Mat):
        dst.view().copy_(src.view()[:, cols.long()])

    def make_seg(self, batch, n_graphs, replicas):
        return torch.cat([batch.int() + r * n_graphs for r in range(replicas)])

    def rownorm(self, x: Mat, rows_per_rep, replicas, y, ld_row, ld_rep):
        v = x.view()
        nrm = v / torch.clamp(v.norm(dim=1, keepdim=True), min=1e-12)
        flat = y.view(-1)
        C = x.cols
        for r in range(replicas):
            blk = nrm[r * rows_per_rep:(r + 1) * rows_per_rep]
            idx = (torch.arange(rows_per_rep)[:, None] * ld_row + r * ld_rep + torch.arange(C)[None, :]).reshape(-1)
            flat[idx] = blk.reshape(-1)

    def cls_attention(self, x, g, cls, y: Mat):
        n, T, C = x.shape
        tok = torch.cat([cls.view(1, 1, C).expand(n, 1, C), x], dim=1)          # n x (T+1) x C
        outs = []
        for h in range(g.shape[0]):
            a = torch.softmax(tok @ g[h], dim=1)                                # n x (T+1)
            outs.append((a[:, :, None] * tok).sum(1))
        y.view().copy_(torch.cat(outs, dim=1))

    def frame_reduce(self, x, mode, y: Mat):
        y.view().copy_(x.mean(1) if mode == "mean" else x.max(1)[0])


# ------------------------------------------------------------------------------------------------
# CorrNet point-branch ops (emulated with the oracle's primitive restatements)
# ------------------------------------------------------------------------------------------------
from oracle import pyg_primitives as _P      # noqa: E402  (tests may import the oracle)


def _batch_from_ptr(ptr):
    p = ptr.long().tolist()
    return torch.cat([torch.full((p[i + 1] - p[i],), i, dtype=torch.long) for i in range(len(p) - 1)])


def _emu_edge_hidden(self, A: Mat, B: Mat, csr: CSR, ec, Z: Mat):
    H = ec.H
    E = int(csr.rowptr[-1])
    src, dst = csr.src[:E].long(), csr.dst[:E].long()
    a = A.base[A.row0 + dst, A.col0:A.col0 + H]
    b = B.base[B.row0 + src, B.col0:B.col0 + H]
    assert not torch.isnan(a).any() and not torch.isnan(b).any()
    h1 = torch.relu(a + b)
    if ec.s1 is not None:
        h1 = h1 * ec.s1[:H] + ec.t1[:H]
    Z.view()[:E] = torch.relu(h1 @ ec.W2[:H, :H].t() + ec.b2[:H]) * ec.s2[:H] + ec.t2[:H]


def _emu_segmax_gemm(self, X: Mat, lin, relu, csr: CSR, out: Mat):
    E = int(csr.rowptr[-1])
    x = X.view()[:E]
    assert not torch.isnan(x).any()
    z = x @ lin.W[: lin.N, : lin.K].t() + lin.bias[: lin.N]
    if relu:
        z = torch.relu(z)
    if lin.scale is not None:
        z = z * lin.scale[: lin.N] + lin.shift[: lin.N]
    res = torch.full((csr.n_nodes, lin.N), float("-inf"))
    res = res.scatter_reduce(0, csr.dst[:E].long()[:, None].expand(-1, lin.N), z, reduce="amax", include_self=True)
    out.view().copy_(res)


def _emu_pointconv_can_fuse(self, pk, max_nbrs):
    return bool(self.emulate_split and pk.get("fused") is not None and max_nbrs == 64 and pk["edge"].s1 is None)


def _emu_pointconv_fused(self, A: Mat, B: Mat, coo, max_nbrs, pk, out: Mat):
    """the definition morig_pointconv_fused implements, written from the slot table (not through the CSR emulation)"""
    ec, l3 = pk["edge"], pk["fused"]
    H, M = ec.H, A.rows
    assert coo.shape == (2, M * max_nbrs) and A.cols == B.cols == H and out.cols == l3.N and out.rows == M
    slots = coo[0].view(M, max_nbrs)
    c = torch.arange(M)[:, None]
    assert int(slots.max()) < B.rows
    keep = (slots >= 0) & (slots != c)                       # remove_self_loops on the raw index pair ...
    src = torch.cat([torch.where(keep, slots, c.expand_as(slots)), c], 1)      # ... add_self_loops: (c, c); fillers are duplicates of it
    a, b = A.view()[:, None, :], B.view()[src]
    assert not torch.isnan(a).any() and not torch.isnan(b).any()
    z2 = torch.relu(torch.relu(a + b) @ ec.W2[:H, :H].t() + ec.b2[:H])
    z3 = torch.relu(z2 @ l3.W[: l3.N, :H].t() + l3.bias[: l3.N])          # BN2 is folded into l3.W / l3.bias (packing.pack_pointconv)
    if l3.scale is not None:
        z3 = z3 * l3.scale[: l3.N] + l3.shift[: l3.N]
    out.view().copy_(z3.amax(1))


def _emu_fps(self, pos: Mat, ptr, out_ptr, start, n_clouds, max_cloud_points, n_samples):
    p = pos.view()[:, :3]
    pp, oo = ptr.long().tolist(), out_ptr.long().tolist()
    out = []
    for b in range(n_clouds):
        n, m = pp[b + 1] - pp[b], oo[b + 1] - oo[b]
        pts = p[pp[b]:pp[b + 1]]
        cur = int(start[b]) if start is not None else 0
        dist = torch.full((n,), float("inf"))
        for _ in range(m):
            out.append(cur + pp[b])
            dist = torch.minimum(dist, ((pts - pts[cur]) ** 2).sum(-1))
            cur = int(torch.argmax(dist))
    assert len(out) == n_samples
    return torch.tensor(out, dtype=torch.int32)


def _emu_ball_query(self, x: Mat, ptr_x, y: Mat, ptr_y, n_clouds, radius, max_nbrs):
    row, col = _P.radius(x.view()[:, :3], y.view()[:, :3], radius, _batch_from_ptr(ptr_x), _batch_from_ptr(ptr_y),
                         max_num_neighbors=max_nbrs)
    M = y.rows
    coo = torch.full((2, M * max_nbrs), -1, dtype=torch.int64)
    slot = torch.zeros(M, dtype=torch.long)
    for r, c in zip(row.tolist(), col.tolist()):
        coo[0, r * max_nbrs + slot[r]] = c
        coo[1, r * max_nbrs + slot[r]] = r
        slot[r] += 1
    return coo


def _emu_knn_interpolate(self, feat: Mat, pos_x: Mat, ptr_x, pos_y: Mat, ptr_y, n_clouds, max_targets_per_cloud, k, out: Mat):
    res = _P.knn_interpolate(feat.view(), pos_x.view()[:, :3], pos_y.view()[:, :3], _batch_from_ptr(ptr_x),
                             _batch_from_ptr(ptr_y), k=k)
    out.view().copy_(res)


def _emu_knn_search(self, pos_x: Mat, ptr_x, pos_y: Mat, ptr_y, n_clouds, max_targets_per_cloud, k):
    """-> (idx [ny, 3] int32, -1 padded; wgt [ny, 3] = 1 / clamp(d^2, 1e-16)): the contract of morig_knn_search"""
    px, py = pos_x.view()[:, :3], pos_y.view()[:, :3]
    yi, xi = _P.knn(px, py, k, batch_x=_batch_from_ptr(ptr_x), batch_y=_batch_from_ptr(ptr_y))
    ny = py.shape[0]
    idx = torch.full((ny, 3), -1, dtype=torch.int32)
    wgt = torch.zeros((ny, 3))
    slot = torch.zeros(ny, dtype=torch.long)
    diff = px[xi] - py[yi]
    w = 1.0 / torch.clamp((diff * diff).sum(-1), min=1e-16)
    for e in range(yi.numel()):                                   # knn lists a target's neighbours consecutively, nearest first
        t = int(yi[e])
        idx[t, slot[t]] = int(xi[e]); wgt[t, slot[t]] = w[e]; slot[t] += 1
    return idx, wgt


def _emu_knn_apply(self, feat: Mat, nn, out: Mat):
    idx, wgt = nn
    f = feat.view()
    w = torch.where(idx >= 0, wgt, torch.zeros_like(wgt)).unsqueeze(-1)
    g = f[idx.clamp(min=0).reshape(-1).long()].reshape(idx.shape[0], idx.shape[1], -1)
    num = torch.zeros((idx.shape[0], f.shape[1]))
    den = torch.zeros((idx.shape[0], 1))
    for s_ in range(idx.shape[1]):                                # scatter_add's order: slot by slot
        num = num + g[:, s_] * w[:, s_]
        den = den + w[:, s_]
    out.view().copy_(num / den)


def _emu_cosine_nn(self, v: Mat, ptr_v, p: Mat, ptr_p, n_clouds, max_rows_per_cloud):
    vv, pp = v.view(), p.view()
    pv, pq = ptr_v.long().tolist(), ptr_p.long().tolist()
    nn = torch.empty(v.rows, dtype=torch.int32)
    sim = torch.empty(v.rows)
    for c in range(n_clouds):
        s = vv[pv[c]:pv[c + 1]] @ pp[pq[c]:pq[c + 1]].t()
        m, i = s.max(dim=1)
        nn[pv[c]:pv[c + 1]] = (i + pq[c]).int()
        sim[pv[c]:pv[c + 1]] = m
    return nn, sim


def _emu_sigmoid_minmax(self, x: Mat, ptr, n_meshes, out: Mat):
    s = torch.sigmoid(x.view())
    p = ptr.long().tolist()
    o = out.view()
    for b in range(n_meshes):
        m = s[p[b]:p[b + 1]]
        o[p[b]:p[b + 1]] = (m - m.min()) / (m.max() - m.min())


def _emu_cosine_knn(self, y: Mat, ptr_y, x: Mat, ptr_x, n_clouds, max_rows_per_cloud, k, vis: Mat = None, split=False):
    yy, xx = y.view(), x.view()
    py, px = ptr_y.long().tolist(), (ptr_y if split else ptr_x).long().tolist()
    idx = torch.full((y.rows, k), -1, dtype=torch.int32)
    vv = vis.view().reshape(-1) if vis is not None else None
    for c in range(n_clouds):
        ys, ye, xs, xe = py[c], py[c + 1], px[c], px[c + 1]
        if ye == ys or xe == xs:
            continue
        s = yy[ys:ye] @ xx[xs:xe].t()
        if split:
            s = s.masked_fill(~(vv[xs:xe] >= 0.5)[None, :], float("-inf"))
        order = torch.sort(s, dim=1, descending=True, stable=True)
        kk = min(k, xe - xs)
        sel = order[1][:, :kk] + xs
        sel = torch.where(torch.isinf(order[0][:, :kk]), torch.full_like(sel, -1), sel)
        if split:
            sel = torch.where((vv[ys:ye] < 0.5)[:, None], sel, torch.full_like(sel, -1))
        idx[ys:ye, :kk] = sel.int()
    return idx


def _emu_flow_vote(self, mode, idx, feat_q: Mat, feat_s: Mat, pos_q, pos_s, vis: Mat, l1: Mat):
    fq, fs, vv, out = feat_q.view(), feat_s.view(), vis.view().reshape(-1), l1.view()
    n, k = idx.shape
    j = idx.long().clamp(min=0)
    ok = (idx >= 0).float()
    dot = (fs[j] * fq[:, None, :]).sum(-1)                       # [n, k]
    if mode == 0:
        w = dot * vv[:, None] * ok
        val = pos_s.view()[j] - pos_q.view()[:, None, :]
        rows = torch.ones(n, dtype=torch.bool)
    else:
        w = dot * ok
        val = out[:, :3][j]
        rows = vv < 0.5
    val = torch.where(ok[..., None] > 0, val, torch.zeros_like(val))      # skipped neighbours add nothing (not NaN * 0)
    flow = (val * w[..., None]).sum(1) / w.sum(1, keepdim=True)
    out[rows, :3] = flow[rows]
    if mode == 0:
        out[:, 3] = vv


def _emu_gather_rows(self, src: Mat, idx, dst: Mat):
    dst.view().copy_(src.view()[idx.long()])


EmuOps.edge_hidden = _emu_edge_hidden
EmuOps.segmax_gemm = _emu_segmax_gemm
EmuOps.pointconv_can_fuse = _emu_pointconv_can_fuse
EmuOps.pointconv_fused = _emu_pointconv_fused
EmuOps.fps = _emu_fps
EmuOps.ball_query = _emu_ball_query
EmuOps.knn_interpolate = _emu_knn_interpolate
EmuOps.knn_search = _emu_knn_search
EmuOps.knn_apply = _emu_knn_apply
EmuOps.cosine_nn = _emu_cosine_nn
EmuOps.gather_rows = _emu_gather_rows
EmuOps.sigmoid_minmax = _emu_sigmoid_minmax
EmuOps.cosine_knn = _emu_cosine_knn
EmuOps.flow_vote = _emu_flow_vote


# ---- joint extraction (csrc/joints.hip): numpy float64, same operation order as the kernels -------------------------
def _np64(t):
    return t.detach().cpu().numpy().astype("float64")


def _emu_inside_mask(self, pts, vox88, translate, scale, dims0):
    import numpy as np
    p = _np64(pts)
    vc = np.round((p - np.asarray(translate, dtype=np.float64)) / float(scale) * float(dims0)).astype(np.int64)
    ok = np.logical_and(np.all(vc >= 0, axis=1), np.all(vc < 88, axis=1))
    vc = np.clip(vc, 0, 87)
    v = vox88.cpu().numpy().reshape(88, 88, 88)
    return torch.from_numpy(np.logical_and(ok, v[vc[:, 0], vc[:, 1], vc[:, 2]] != 0))


def _emu_knn_bandwidth(self, pts, k):
    import numpy as np
    p = _np64(pts)
    d2 = ((p[:, None, :] - p[None, :, :]) ** 2).sum(-1)
    kth = np.sqrt(np.partition(d2, k - 1, axis=1)[:, k - 1])
    return torch.tensor([kth.sum() / len(p)], dtype=torch.float64)


def _emu_meanshift(self, pts, weights, bandwidth, max_iter):
    import numpy as np
    p = _np64(pts)
    h2 = float(bandwidth.item()) ** 2
    w = None if weights is None else weights.cpu().numpy().astype(np.float64).reshape(-1)
    diff2, t = 1e20, 1
    while np.sqrt(diff2) > 1e-3 and t < max_iter:
        d2 = ((p[:, None, :] - p[None, :, :]) ** 2).sum(-1)            # [target j, source i]
        k = np.maximum(h2 - d2, 0.0)
        if w is not None:
            k = k * w[None, :]
        moved = 0.3 * ((k @ p) / (k.sum(1, keepdims=True) + 1e-10) - p) + p
        diff2 = ((moved - p) ** 2).sum()
        p = moved
        t += 1
    return torch.from_numpy(p)


def _emu_nms_counts(self, pts, bandwidth):
    import numpy as np
    p = _np64(pts)
    d = np.sqrt(((p[:, None, :] - p[None, :, :]) ** 2).sum(-1))
    return torch.from_numpy((d <= float(bandwidth.item())).sum(0).astype(np.int32))


def _emu_nms_greedy(self, pts, attn, bandwidth, order, thrd_density, thrd_attn):
    import numpy as np
    p = _np64(pts)
    a = attn.cpu().numpy().astype(np.float32).reshape(-1)
    n = len(p)
    d = np.sqrt(((p[:, None, :] - p[None, :, :]) ** 2).sum(-1))
    h = float(bandwidth.item())
    alive = np.ones(n, dtype=bool)
    for i in order.cpu().numpy().tolist():
        if alive[i]:
            nb = d[:, i] <= h
            alive[nb] = False
            if a[nb].max() > np.float32(thrd_attn) or nb.sum() / n > thrd_density:
                alive[i] = True
    return torch.from_numpy(alive)


# batched forms: mesh b = rows [ptr[b], ptr[b + 1]); per mesh exactly the one-set emulation
def _ranges(ptr):
    p = ptr.tolist()
    return [(p[b], p[b + 1]) for b in range(len(p) - 1)]


def _emu_knn_bandwidth_batched(self, pts, ptr, max_n, quantile):
    return torch.cat([_emu_knn_bandwidth(self, pts[s:e], max(int((e - s) * quantile), 1)) if e > s else torch.zeros(1, dtype=torch.float64)
                      for s, e in _ranges(ptr)])


def _emu_meanshift_batched(self, pts, weights, ptr, max_n, bandwidth, max_iter):
    out = torch.empty_like(pts)
    for b, (s, e) in enumerate(_ranges(ptr)):
        if e > s:
            out[s:e] = _emu_meanshift(self, pts[s:e], None if weights is None else weights[s:e], bandwidth[b:b + 1], max_iter)
    return out


def _emu_nms_counts_batched(self, pts, ptr, max_n, bandwidth):
    out = torch.zeros(pts.shape[0], dtype=torch.int32)
    for b, (s, e) in enumerate(_ranges(ptr)):
        if e > s:
            out[s:e] = _emu_nms_counts(self, pts[s:e], bandwidth[b:b + 1])
    return out


def _emu_nms_greedy_batched(self, pts, attn, ptr, bandwidth, order_local, thrd_density, thrd_attn):
    out = torch.zeros(pts.shape[0], dtype=torch.bool)
    for b, (s, e) in enumerate(_ranges(ptr)):
        if e > s:
            out[s:e] = _emu_nms_greedy(self, pts[s:e], attn[s:e], bandwidth[b:b + 1], order_local[s:e], thrd_density, thrd_attn)
    return out


EmuOps.knn_bandwidth_batched = _emu_knn_bandwidth_batched
EmuOps.meanshift_batched = _emu_meanshift_batched


def _emu_meanshift_batched_sorted(self, pts, weights, ptr, max_n, bandwidth, max_iter, with_counts=False):
    modes = _emu_meanshift_batched(self, pts, weights, ptr, max_n, bandwidth, max_iter)
    return (modes, _emu_nms_counts_batched(self, modes, ptr, max_n, bandwidth)) if with_counts else modes


EmuOps.meanshift_batched_sorted = _emu_meanshift_batched_sorted
EmuOps.nms_counts_batched = _emu_nms_counts_batched
EmuOps.nms_greedy_batched = _emu_nms_greedy_batched
EmuOps.inside_mask = _emu_inside_mask
EmuOps.knn_bandwidth = _emu_knn_bandwidth
EmuOps.meanshift = _emu_meanshift
EmuOps.nms_counts = _emu_nms_counts
EmuOps.nms_greedy = _emu_nms_greedy


def _emu_radius_sample(self, x: Mat, y: Mat, radius, max_nbrs, seed):
    """morig_radius_sample: inclusive ball, uniform random subset of max_nbrs for over-full rows (emulation: torch.randperm)."""
    xv, yv = x.view()[:, :3], y.view()[:, :3]
    r2 = float(torch.tensor(radius, dtype=torch.float64) ** 2)
    d2 = ((yv[:, None, :] - xv[None, :, :]) ** 2).sum(-1)
    valid = d2 <= torch.tensor(r2, dtype=torch.float32)
    ny = yv.shape[0]
    coo = torch.full((2, ny * max_nbrs), -1, dtype=torch.int64)
    g = torch.Generator().manual_seed(int(seed))
    for i in range(ny):
        hits = torch.nonzero(valid[i]).flatten()
        if hits.numel() > max_nbrs:
            hits = hits[torch.randperm(hits.numel(), generator=g)[:max_nbrs]]
        coo[0, i * max_nbrs:i * max_nbrs + hits.numel()] = hits
        coo[1, i * max_nbrs:i * max_nbrs + hits.numel()] = i
    return coo, valid.sum(1).to(torch.int32)


EmuOps.radius_sample = _emu_radius_sample


# ------------------------------------------------------------------------------------------------
# train-mode forward support (csrc/train_ops.hip)
# ------------------------------------------------------------------------------------------------
def _emu_col_stats(self, X: Mat, rows_dev=None):
    rows = int(rows_dev.item()) if rows_dev is not None else X.rows
    x = X.view()[:rows].double()
    assert not torch.isnan(x).any()
    mean = x.mean(0)
    var = (x * x).mean(0) - mean * mean
    return mean.float(), var.clamp(min=0).float(), torch.tensor([float(rows)])


def _emu_col_affine(self, X: Mat, scale, shift, rows_dev=None, out=None):
    rows = int(rows_dev.item()) if rows_dev is not None else X.rows
    v = X.view()
    (out if out is not None else X).view()[:rows] = v[:rows] * scale[: X.cols] + shift[: X.cols]


def _emu_edge_gather_relu(self, A: Mat, B: Mat, csr: CSR, Z: Mat, want_stats=False):
    E = int(csr.rowptr[-1])
    src, dst = csr.src[:E].long(), csr.dst[:E].long()
    Z.view()[:E] = torch.relu(A.view()[dst] + B.view()[src])
    return self.col_stats(Z, rows_dev=csr.rowptr[-1:]) if want_stats else None


def _emu_segmax_affine(self, Z: Mat, rowptr, n_segments, out: Mat, scale=None, shift=None):
    p = rowptr.long().tolist()
    z = Z.view()
    if scale is not None:
        z = z * scale[: Z.cols] + shift[: Z.cols]
    res = torch.zeros((n_segments, Z.cols))
    for v in range(n_segments):
        if p[v + 1] > p[v]:
            assert not torch.isnan(z[p[v]:p[v + 1]]).any()
            res[v] = z[p[v]:p[v + 1]].max(0)[0]
    out.view().copy_(res)


# ---- train-mode backward operators (csrc/train_bwd.hip): the formulas, in float64 ----------------------------------------------
def _rows(M: Mat, rows_dev):
    return int(rows_dev.item()) if rows_dev is not None else M.rows


def _emu_bn_backward_stats(self, dz: Mat, y=None, mean=None, rstd=None, rows_dev=None):
    r = _rows(dz, rows_dev)
    g = dz.view()[:r].double()
    assert not torch.isnan(g).any(), ("uninitialised rows read", tuple(g.shape), torch.nonzero(torch.isnan(g).any(1)).flatten()[:8].tolist())
    if y is None:
        return g.sum(0).float(), None
    xh = (y.view()[:r].double() - mean.double()) * rstd.double()
    return g.sum(0).float(), (g * xh).sum(0).float()


def _emu_bn_relu_backward(self, dz: Mat, y: Mat, mean, rstd, gamma, sum_dz, sum_dzx, du: Mat, rows_dev=None, want_sum=False):
    r = _rows(dz, rows_dev)
    yv = y.view()[:r]
    xh = (yv - mean) * rstd
    g = gamma * rstd * (dz.view()[:r] - sum_dz / r - xh * (sum_dzx / r))
    du.view()[:r] = torch.where(yv > 0, g, torch.zeros_like(g))
    return du.view()[:r].double().sum(0).float() if want_sum else None


def _emu_segmax_affine_arg(self, Z: Mat, rowptr, n_segments, out: Mat, scale=None, shift=None, want_zwin=False):
    p = rowptr.long().tolist()
    z = z_raw = Z.view()
    if scale is not None:
        z = z * scale[: Z.cols] + shift[: Z.cols]
    res = torch.zeros((n_segments, Z.cols))
    arg = torch.full((n_segments, Z.cols), -1, dtype=torch.int32)
    pt = rowptr.long()[: n_segments + 1]
    lens = pt[1:] - pt[:-1]
    lmax = int(lens.max()) if n_segments else 0
    if n_segments > 64 and 0 < lmax and n_segments * lmax * Z.cols <= 2 ** 28:
        # many short segments: one padded [segments, longest, columns] gather instead of a Python loop over the segments (the
        # 40 000-segment case took 33 s of the GPU suite's host time this way); same first-maximum rule
        k = torch.arange(lmax)
        valid = k[None, :] < lens[:, None]
        zz = z[(pt[:-1, None] + k[None, :]).clamp(max=z.shape[0] - 1)]
        zz = torch.where(valid[:, :, None], zz, torch.full((), float("-inf")))
        top = zz.max(1)[0]
        first = (zz == top[:, None, :]).int().argmax(1)
        live = (lens > 0)[:, None]
        res = torch.where(live, top, torch.zeros(()))
        arg = torch.where(live, first + pt[:-1, None], torch.full((), -1, dtype=torch.long)).int()
    else:
        for v in range(n_segments):
            if p[v + 1] > p[v]:
                seg = z[p[v]:p[v + 1]]
                res[v] = seg.max(0)[0]
                arg[v] = ((seg == res[v]).int().argmax(0) + p[v]).int()      # first maximum
    out.view().copy_(res)
    if want_zwin:
        cc = torch.arange(Z.cols).expand_as(arg)
        zwin = torch.where(arg >= 0, z_raw[arg.clamp(min=0).long(), cc], torch.zeros(()))
        return arg, zwin.contiguous()
    return arg


def _dense_dz(dout: Mat, arg, rows, cols):
    dz = torch.zeros((rows, cols), dtype=torch.float64)
    live = arg >= 0
    cc = torch.arange(cols).expand_as(arg)
    dz[arg[live].long(), cc[live]] = dout.view().double()[live]
    return dz


def _emu_segmax_bn_backward_stats(self, dout: Mat, arg, Z, mean, rstd, zwin=None):
    # as the kernel: only the winning rows of Z are read (rows past the live edge count may hold anything)
    live = arg >= 0
    cols = torch.arange(arg.shape[1]).expand_as(arg)
    g = torch.where(live, dout.view().double(), torch.zeros((), dtype=torch.float64))
    zwin = zwin.double() if zwin is not None else Z.view().double()[arg.clamp(min=0).long(), cols]
    assert not torch.isnan(zwin[live]).any()
    xh = torch.where(live, (zwin - mean.double()) * rstd.double(), torch.zeros((), dtype=torch.float64))
    return g.sum(0).float(), (g * xh).sum(0).float()


def _emu_segmax_bn_relu_backward(self, dout: Mat, arg, Z: Mat, rowptr, seg_of_row, mean, rstd, gamma, sum_dz, sum_dzx, du: Mat, relu=True,
                                 want_sum=False):
    r = int(rowptr[-1])
    dz = _dense_dz(dout, arg, Z.rows, Z.cols)[:r].float()
    zv = Z.view()[:r]
    xh = (zv - mean) * rstd
    g = gamma * rstd * (dz - sum_dz / r - xh * (sum_dzx / r))
    du.view()[:r] = torch.where(zv > 0, g, torch.zeros_like(g)) if relu else g
    du.view()[r:] = 0.0                                 # the operator clears the rows past the live count
    return du.view()[:r].double().sum(0).float() if want_sum else None


def _emu_edge_scatter_backward(self, dG: Mat, csr: CSR, n_src, dA: Mat, dB: Mat):
    E = int(csr.rowptr[-1])
    g = dG.view()[:E]
    a = torch.zeros((csr.n_nodes, dG.cols)).index_add_(0, csr.dst[:E].long(), g)
    b = torch.zeros((n_src, dG.cols)).index_add_(0, csr.src[:E].long(), g)
    dA.view().copy_(a)
    dB.view().copy_(b)


def _emu_edge_bn_scatter_backward(self, dG: Mat, Y, csr: CSR, n_src, dA: Mat, dB: Mat, mean=None, rstd=None, gamma=None, sum_dz=None,
                                  sum_dzx=None, ZA=None, ZB=None):
    E = int(csr.rowptr[-1])
    g = dG.view()[:E]
    if mean is not None:
        yv = Y.view()[:E] if ZA is None else torch.relu(ZA.view()[csr.dst[:E].long()] + ZB.view()[csr.src[:E].long()])
        xh = (yv - mean) * rstd
        d = gamma * rstd * (g - sum_dz / E - xh * (sum_dzx / E))
        g = torch.where(yv > 0, d, torch.zeros_like(d))
    rowptr_t, perm_t = csr.transposed(n_src)            # (exercises the transposed graph the kernel walks)
    assert int(rowptr_t[-1]) == E and torch.equal(torch.sort(perm_t[:E].long())[0], torch.arange(E))
    a = torch.zeros((csr.n_nodes, dG.cols)).index_add_(0, csr.dst[:E].long(), g)
    seg = torch.repeat_interleave(torch.arange(n_src), (rowptr_t[1:] - rowptr_t[:-1]).long())
    assert torch.equal(csr.src[:E].long()[perm_t[:E].long()], seg)
    b = torch.zeros((n_src, dG.cols)).index_add_(0, seg, g[perm_t[:E].long()])
    dA.view().copy_(a)
    dB.view().copy_(b)


def _emu_edge_bn_sums_from_products(self, M, db2, W2, mean, rstd):
    h_out, h_in = W2.shape
    w, m = W2.double(), M.double()
    a = db2[:h_out].double() @ w
    b = (w * m).sum(0)
    return a.float(), (rstd[:h_in].double() * (b - mean[:h_in].double() * a)).float()


def _emu_gemm_tn(self, A: Mat, B: Mat, out=None, rows_dev=None, b_shift=None):
    r = _rows(A, rows_dev)
    b = B.view()[:r].double()
    if b_shift is not None:                                  # rows of B centred on b_shift first (morig_gemm_tn_shift)
        b = b - b_shift[:B.cols].double()[None, :]
    res = (A.view()[:r].double().t() @ b).float()
    if out is not None:
        out.view().copy_(res)
        return out.base
    return res


def _emu_flag(self, device):
    if not hasattr(self, "_ovf"):
        self._ovf = torch.zeros(1, dtype=torch.int32)
    return self._ovf


EmuOps.bn_backward_stats = _emu_bn_backward_stats
EmuOps.bn_relu_backward = _emu_bn_relu_backward
EmuOps.segmax_affine_arg = _emu_segmax_affine_arg
EmuOps.segmax_bn_backward_stats = _emu_segmax_bn_backward_stats
EmuOps.segmax_bn_relu_backward = _emu_segmax_bn_relu_backward
EmuOps.edge_scatter_backward = _emu_edge_scatter_backward
EmuOps.edge_bn_scatter_backward = _emu_edge_bn_scatter_backward
EmuOps.edge_bn_sums_from_products = _emu_edge_bn_sums_from_products
EmuOps.gemm_tn = _emu_gemm_tn
EmuOps._flag = _emu_flag
EmuOps.col_stats = _emu_col_stats


def _emu_bn_finalize(self, bn, mean, var, count):
    with torch.no_grad():
        sd = torch.sqrt(var + bn.eps)
        s = (bn.weight.detach().float() if bn.weight is not None else torch.ones_like(var)) / sd
        t = (bn.bias.detach().float() if bn.bias is not None else torch.zeros_like(var)) - mean * s
        if bn.track_running_stats and bn.running_mean is not None:
            bn.num_batches_tracked += 1
            m = float(bn.momentum)
            unbiased = var * (count / torch.clamp(count - 1.0, min=1.0))
            bn.running_mean.mul_(1.0 - m).add_(mean, alpha=m)
            bn.running_var.mul_(1.0 - m).add_(unbiased, alpha=m)
    return s.contiguous(), t.contiguous(), (1.0 / sd).contiguous()


EmuOps.bn_finalize = _emu_bn_finalize
EmuOps.col_affine = _emu_col_affine
EmuOps.edge_gather_relu = _emu_edge_gather_relu
EmuOps.segmax_affine = _emu_segmax_affine
