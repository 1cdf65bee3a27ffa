"""Joint extraction after the hot path on the MI355X-native op layer: the functions of
/root/reference/utils/cluster_utils.py (``meanshift_cluster`` :14-38, ``nms_meanshift`` :41-66),
/root/reference/utils/mst_utils.py (``inside_check`` :15-29, ``flip`` :294-313) and sklearn's
``estimate_bandwidth`` as evaluate/eval_rigging.py:89 calls it, with the reference's names and argument meaning, and
``extract_joints`` = the sequence of evaluate/eval_rigging.py:72-95.

The reference computes in numpy float64 on the host; here the O(n^2) scans (k-th nearest neighbour, mean-shift steps,
neighbour counts, the greedy suppression) are float64 HIP kernels (csrc/joints.hip) on device tensors. Point sets are
``torch.float64 [n, 3]`` CUDA tensors (numpy input is uploaded); ``attn`` is ``float32 [n, 1]`` as the reference
loads it from ``*_attn.npy``. There is no CPU fallback: without the HIP library or a GPU the op layer raises.
"""
from __future__ import annotations

from typing import Optional

import os
import numpy as np
import torch

from .runtime import get_ops


def _dev_pts(pts, device) -> torch.Tensor:
    t = torch.as_tensor(pts) if not torch.is_tensor(pts) else pts
    return t.to(device=device, dtype=torch.float64).contiguous()


def _dev_attn(attn, device) -> torch.Tensor:
    t = torch.as_tensor(attn) if not torch.is_tensor(attn) else attn
    return t.to(device=device, dtype=torch.float32).reshape(-1, 1).contiguous()


def inside_check(pts: torch.Tensor, vox):
    """utils/mst_utils.py:15-29. vox: object with ``data`` (88^3 bool), ``translate``, ``scale``, ``dims`` (a binvox
    ``Voxels``). Returns (pts[inside], indices of the inside points)."""
    ops = get_ops()
    data = torch.as_tensor(np.ascontiguousarray(np.asarray(vox.data, dtype=np.uint8))).to(pts.device).reshape(-1)
    keep = ops.inside_mask(pts, data, vox.translate, vox.scale, vox.dims[0])
    return pts[keep], torch.nonzero(keep).squeeze(1)


def estimate_bandwidth(pts: torch.Tensor, quantile: float = 0.3) -> torch.Tensor:
    """sklearn.cluster.estimate_bandwidth(X, quantile): device tensor [1] (float64)."""
    n = pts.shape[0]
    return get_ops().knn_bandwidth(pts, max(int(n * quantile), 1))


def meanshift_cluster(pts_in: torch.Tensor, bandwidth, weights: Optional[torch.Tensor] = None, max_iter: int = 20) -> torch.Tensor:
    """utils/cluster_utils.py:14-38. bandwidth: device tensor [1] or a float."""
    bw = bandwidth if torch.is_tensor(bandwidth) else torch.tensor([float(bandwidth)], dtype=torch.float64, device=pts_in.device)
    w = None if weights is None else weights.reshape(-1).contiguous()
    return get_ops().meanshift(pts_in, w, bw, max_iter)


def nms_meanshift(pts_in: torch.Tensor, attn: torch.Tensor, bandwidth, thrd_density: float, thrd_attn: float = 0.7) -> torch.Tensor:
    """utils/cluster_utils.py:41-66. The visiting order is numpy's ``argsort(counts)[::-1]`` (:52) on the host -- its
    unstable sort decides the order among equal counts -- everything else runs on the device."""
    ops = get_ops()
    bw = bandwidth if torch.is_tensor(bandwidth) else torch.tensor([float(bandwidth)], dtype=torch.float64, device=pts_in.device)
    counts = ops.nms_counts(pts_in, bw)
    order = np.argsort(counts.cpu().numpy().astype(np.int64))[::-1]
    order_d = torch.as_tensor(np.ascontiguousarray(order).astype(np.int32)).to(pts_in.device)
    alive = ops.nms_greedy(pts_in, attn.reshape(-1).contiguous(), bw, order_d, thrd_density, thrd_attn)
    return pts_in[alive]


def flip(pred_joints):
    """utils/mst_utils.py:294-313 on the (small) joint array; returns numpy (joints, side_indicator)."""
    j = pred_joints.detach().cpu().numpy() if torch.is_tensor(pred_joints) else np.asarray(pred_joints)
    left = j[j[:, 0] < -2e-2].reshape(-1, 3)
    middle = j[np.abs(j[:, 0]) <= 2e-2].reshape(-1, 3).copy()
    middle[:, 0] = 0.0
    right = left.copy()
    right[:, 0] = -right[:, 0]
    side = np.concatenate((-np.ones(len(left)), np.zeros(len(middle)), np.ones(len(right))), axis=0)
    return np.concatenate((left, middle, right), axis=0), side


def extract_joints(shifted_pts, attn, vox=None, bandwidth_quantile: float = 0.04, threshold1: float = 0.1,
                   threshold2: float = 0.02, max_iter: int = 30, device=None):
    """evaluate/eval_rigging.py:72-95 from the loaded arrays on: attention min-max normalised (:72), inside test (:80),
    attention threshold (:82-83), x-mirror (:86-88), bandwidth (:89), mean-shift (:91), NMS (:94), flip (:95).
    Returns dict(joints numpy [J, 3], side, bandwidth float, modes device [2m, 3], attn device [2m, 1])."""
    if device is None:
        device = shifted_pts.device if torch.is_tensor(shifted_pts) and shifted_pts.is_cuda else torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    if device.type == "cuda" and torch.cuda.current_device() != device.index:
        with torch.cuda.device(device):                                   # the native launches go to the current device's stream
            return extract_joints(shifted_pts, attn, vox, bandwidth_quantile, threshold1, threshold2, max_iter, device)
    pts = _dev_pts(shifted_pts, device)
    a = _dev_attn(attn, device)
    a = (a - a.min()) / (a.max() - a.min())                               # float32, as numpy does on the loaded array
    if vox is not None:
        pts, inside = inside_check(pts, vox)
        a = a[inside, :]
    sel = a.squeeze(1) > threshold1
    pts, a = pts[sel], a[sel]
    mirror = torch.tensor([[-1.0, 1.0, 1.0]], dtype=torch.float64, device=device)
    pts = torch.cat((pts, pts * mirror), dim=0).contiguous()
    a = a.repeat(2, 1).contiguous()
    bw = estimate_bandwidth(pts, bandwidth_quantile)
    modes = meanshift_cluster(pts, bw, a, max_iter=max_iter)
    kept = nms_meanshift(modes, a, bw, threshold2)
    joints, side = flip(kept)
    return dict(joints=joints, side=side, bandwidth=float(bw.item()), modes=modes, attn=a)


_POOL = None


def _sort_pool():
    global _POOL
    if _POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=max(1, min(8, (os.cpu_count() or 2) // 2)))
    return _POOL


def extract_joints_batched(shifted_pts, attn, batch, vox=None, bandwidth_quantile: float = 0.04, threshold1: float = 0.1,
                           threshold2: float = 0.02, max_iter: int = 30, num_graphs: Optional[int] = None):
    """``extract_joints`` for ALL meshes of a batch at once (the reference loops over models, evaluate/eval_rigging.py:62-98):
    shifted_pts [N, 3] / attn [N, 1] of the concatenated meshes, ``batch`` = PyG's sorted mesh index per vertex, ``vox`` = None or
    one voxel object per mesh. Every O(n^2) stage -- bandwidth, mean-shift steps, neighbour counts, greedy suppression -- is ONE
    launch set over all point sets (segment-aware kernels, csrc/joints.hip); per mesh the arithmetic is that of ``extract_joints``.
    Three host round trips per batch instead of ~10 per mesh: the kept-point counts, the neighbour counts (numpy's
    ``argsort(counts)[::-1]`` of cluster_utils.py:52 fixes the visiting order among equal counts, so that one call stays numpy,
    per mesh) and the survivors. Returns one dict per mesh, as ``extract_joints``."""
    ops = get_ops()
    # device resolution as in extract_joints: a CUDA tensor's own device, else the current one; the native launches go to the
    # CURRENT device's stream, so a tensor on another GPU switches the current device for the call
    from . import runtime
    if runtime._test_ops is not None and torch.is_tensor(shifted_pts) and not shifted_pts.is_cuda:
        device = shifted_pts.device                      # tests/ only: host logic on the CPU emulation of the op layer
    else:
        device = (shifted_pts.device if torch.is_tensor(shifted_pts) and shifted_pts.is_cuda
                  else torch.device("cuda", torch.cuda.current_device()))
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if torch.cuda.current_device() != device.index:
            with torch.cuda.device(device):
                return extract_joints_batched(shifted_pts, attn, batch, vox, bandwidth_quantile, threshold1, threshold2, max_iter, num_graphs)
    pts = _dev_pts(shifted_pts, device)
    a = _dev_attn(attn, device)
    batch = torch.as_tensor(batch).to(device)
    B = int(num_graphs) if num_graphs is not None else (int(batch.max().item()) + 1 if batch.numel() else 0)
    if B == 0 or pts.shape[0] == 0:
        return [dict(joints=np.zeros((0, 3)), side=np.zeros(0), bandwidth=float("nan"),
                     modes=torch.zeros((0, 3), dtype=torch.float64, device=device),
                     attn=torch.zeros((0, 1), dtype=torch.float32, device=device)) for _ in range(B)]
    # attention min-max normalised PER MESH in float32, as numpy does on each loaded array (eval_rigging.py:72). `batch` is sorted
    # (PyG), so the per-mesh extrema are segment reductions (a scatter with 64 target addresses serialises on its atomics: 3 ms each)
    n_per = torch.bincount(batch, minlength=B)
    amin = torch.segment_reduce(a.reshape(-1), "min", lengths=n_per, unsafe=True)[:, None]
    amax = torch.segment_reduce(a.reshape(-1), "max", lengths=n_per, unsafe=True)[:, None]
    # a mesh WITHOUT vertices has an empty segment (+inf / -inf extrema): nothing indexes those rows (`batch` never names the
    # mesh), but they must not stay non-finite in case a later reduction touches the whole column
    empty = (n_per == 0)[:, None]
    amin = torch.where(empty, torch.zeros_like(amin), amin)
    amax = torch.where(empty, torch.ones_like(amax), amax)
    a = (a - amin[batch]) / (amax[batch] - amin[batch])
    keep = a.squeeze(1) > threshold1
    if vox is not None:
        counts_v = n_per.tolist()
        off = 0
        for b, v in enumerate(vox):
            if v is not None:
                data = torch.as_tensor(np.ascontiguousarray(np.asarray(v.data, dtype=np.uint8))).to(device).reshape(-1)
                keep[off:off + counts_v[b]] &= ops.inside_mask(pts[off:off + counts_v[b]].contiguous(), data, v.translate, v.scale, v.dims[0])
            off += counts_v[b]
    kb = batch[keep]
    m = torch.bincount(kb, minlength=B)                                   # kept points per mesh
    m_host = m.tolist()                                                   # host round trip 1
    offs = torch.cumsum(m, 0) - m
    # per mesh [kept ; x-mirrored kept] (eval_rigging.py:86-88), meshes concatenated
    pos_in = torch.arange(kb.numel(), device=device) - offs[kb]
    d1 = 2 * offs[kb] + pos_in
    d2 = d1 + m[kb]
    n2 = 2 * kb.numel()
    P = torch.empty((n2, 3), dtype=torch.float64, device=device)
    A = torch.empty((n2, 1), dtype=torch.float32, device=device)
    pk, ak = pts[keep], a[keep]
    P[d1] = pk
    P[d2] = pk * torch.tensor([[-1.0, 1.0, 1.0]], dtype=torch.float64, device=device)
    A[d1] = ak
    A[d2] = ak
    sizes = [2 * x for x in m_host]
    ptr_host = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    ptr = torch.from_numpy(ptr_host).to(device)
    max_n = max(max(sizes), 1)
    if n2 == 0:
        return [dict(joints=np.zeros((0, 3)), side=np.zeros(0), bandwidth=float("nan"), modes=P, attn=A) for _ in range(B)]
    bw = ops.knn_bandwidth_batched(P, ptr, max_n, bandwidth_quantile)
    modes, counts_d = ops.meanshift_batched_sorted(P, A.reshape(-1).contiguous(), ptr, max_n, bw, max_iter, with_counts=True)
    counts = counts_d.cpu().numpy().astype(np.int64)                        # host round trip 2
    order = np.empty(n2, dtype=np.int32)

    def _order(b):
        s, e = int(ptr_host[b]), int(ptr_host[b + 1])
        order[s:e] = np.argsort(counts[s:e])[::-1]                        # cluster_utils.py:52, per mesh, local indices

    if B >= 8:                                                            # numpy's sort releases the GIL: the meshes sort side by side
        list(_sort_pool().map(_order, range(B)))
    else:
        for b in range(B):
            _order(b)
    alive = ops.nms_greedy_batched(modes, A.reshape(-1).contiguous(), ptr, bw, torch.from_numpy(order).to(device), threshold2, 0.7)
    # host round trip 3: only the survivors travel (a few dozen rows per mesh), with their mesh index
    mesh_of = torch.repeat_interleave(torch.arange(B, device=device), torch.as_tensor(sizes, device=device))
    kept_h = torch.cat([modes[alive], mesh_of[alive].to(torch.float64)[:, None], bw[mesh_of[alive]][:, None]], dim=1).cpu().numpy()
    bw_h = bw.cpu().numpy()
    kept_mesh = kept_h[:, 3].astype(np.int64)
    out = []
    for b in range(B):
        s, e = int(ptr_host[b]), int(ptr_host[b + 1])
        if e == s:
            out.append(dict(joints=np.zeros((0, 3)), side=np.zeros(0), bandwidth=float("nan"), modes=modes[s:e], attn=A[s:e]))
            continue
        joints, side = flip(kept_h[kept_mesh == b, :3])
        out.append(dict(joints=joints, side=side, bandwidth=float(bw_h[b]), modes=modes[s:e], attn=A[s:e]))
    return out
