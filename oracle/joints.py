"""TEST INFRASTRUCTURE ONLY -- never imported by the product (``morig_amd``).

CPU oracle for the joint extraction that follows the hot path: a plain numpy float64 restatement of

    utils/mst_utils.py:15-29      inside_check        (voxel inside-test of the shifted vertices)
    evaluate/eval_rigging.py:80-95  the extraction sequence: attention threshold, x-mirror, bandwidth, mean-shift,
                                    non-maximum suppression, flip
    sklearn.cluster.estimate_bandwidth(X, quantile)   (third party, scikit-learn 1.x: mean distance to the
                                    int(n * quantile)-th nearest neighbour, the point itself included)
    utils/cluster_utils.py:14-38  meanshift_cluster
    utils/cluster_utils.py:41-66  nms_meanshift
    utils/mst_utils.py:294-313    flip

It is pinned against the reference's own functions (imported where they lie, oracle/make_golden.py) through
tests/golden/joints_*.npz. Allowed importers: tests/.
"""
from __future__ import annotations

import numpy as np


def inside_check(pts: np.ndarray, vox_data: np.ndarray, translate, scale: float, dims) -> tuple:
    """mst_utils.py:15-29. ``np.round`` rounds half to even; the bound 88 is hard-coded in the reference (:24-25)."""
    vc = (pts - np.asarray(translate)) / scale * dims[0]
    vc = np.round(vc).astype(int)
    in_grid = np.logical_and(np.all(vc >= 0, axis=1), np.all(vc < 88, axis=1))
    vc = np.clip(vc, 0, 87)
    filled = vox_data[vc[:, 0], vc[:, 1], vc[:, 2]]
    keep = np.logical_and(in_grid, filled)
    return pts[keep], np.argwhere(keep).squeeze()


def estimate_bandwidth(x: np.ndarray, quantile: float) -> float:
    """sklearn.cluster.estimate_bandwidth: k = int(n * quantile) (at least 1) nearest neighbours *including the point
    itself*; the bandwidth is the mean over points of the distance to the k-th of them."""
    n = x.shape[0]
    k = max(int(n * quantile), 1)
    d = np.sqrt(((x[:, None, :] - x[None, :, :]) ** 2).sum(-1))
    kth = np.partition(d, k - 1, axis=1)[:, k - 1]
    return float(kth.sum() / n)


def meanshift_cluster(pts: np.ndarray, bandwidth: float, weights=None, max_iter: int = 20) -> np.ndarray:
    """cluster_utils.py:14-38: Epanechnikov-profile weights max(h^2 - d^2, 0) (times the source's attention), each point
    moves 0.3 of the way to its weighted mean; stops when the total displacement drops to 1e-3 or after max_iter - 1 steps."""
    diff, it = 1e10, 1
    while diff > 1e-3 and it < max_iter:
        d2 = ((pts[None, :, :] - pts[:, None, :]) ** 2).sum(2)              # [source i, target j]
        k = np.maximum(bandwidth ** 2 - d2, 0.0)
        if weights is not None:
            k = k * weights                                                # weights [n, 1]: scales row (source) i
        col = k.sum(axis=0, keepdims=True)
        p = (k / (col + 1e-10)).T
        moved = 0.3 * (p @ pts - pts) + pts
        diff = np.sqrt(((moved - pts) ** 2).sum())
        pts = moved
        it += 1
    return pts


def nms_order(num_neighbors: np.ndarray) -> np.ndarray:
    """the visiting order of nms_meanshift (cluster_utils.py:52): ``np.argsort(counts)[::-1]`` -- numpy's default
    (unstable) sort decides the order among equal counts, so product and oracle both take it from numpy."""
    return np.argsort(num_neighbors)[::-1]


def nms_meanshift(pts: np.ndarray, attn: np.ndarray, bandwidth: float, thrd_density: float, thrd_attn: float = 0.7):
    """cluster_utils.py:41-66: visit points by decreasing neighbour count; a point still marked suppresses every point
    within ``bandwidth`` (itself included) and survives only if its neighbourhood is dense or well attended."""
    n = len(pts)
    dist = np.sqrt(((pts[None, :, :] - pts[:, None, :]) ** 2).sum(2))
    counts = (dist <= bandwidth).sum(axis=0)
    alive = np.ones(n, dtype=bool)
    for i in nms_order(counts):
        if alive[i]:
            nbrs = np.argwhere(dist[:, i] <= bandwidth).squeeze(axis=1)
            attn_max = attn[nbrs].max()
            density = len(nbrs) / n
            alive[nbrs] = False
            if attn_max > thrd_attn or density > thrd_density:
                alive[i] = True
    return pts[alive], alive, counts


def flip(joints: np.ndarray):
    """mst_utils.py:294-313: keep the left half space (x < -0.02), snap the middle band to x = 0, mirror left to right."""
    left = joints[joints[:, 0] < -2e-2].reshape(-1, 3)
    mid = joints[np.abs(joints[:, 0]) <= 2e-2].reshape(-1, 3).copy()
    mid[:, 0] = 0.0
    right = left.copy()
    right[:, 0] = -right[:, 0]
    side = np.concatenate([-np.ones(len(left)), np.zeros(len(mid)), np.ones(len(right))])
    return np.concatenate([left, mid, right], axis=0), side


def extract_joints(shifted_pts: np.ndarray, attn: np.ndarray, vox=None, bandwidth_quantile: float = 0.04,
                   threshold1: float = 0.1, threshold2: float = 0.02, max_iter: int = 30):
    """evaluate/eval_rigging.py:72-95 from the loaded arrays on: attn min-max normalised (:72), inside test (:80),
    attention threshold (:82-83), x-mirror (:86-88), bandwidth (:89), mean-shift (:91), NMS (:94), flip (:95).
    vox: None or (data[88,88,88] bool, translate[3], scale, dims)."""
    attn = (attn - np.min(attn)) / (np.max(attn) - np.min(attn))
    if vox is not None:
        shifted_pts, inside = inside_check(shifted_pts, *vox)
        attn = attn[inside, :]
    sel = attn.squeeze() > threshold1
    shifted_pts, attn = shifted_pts[sel], attn[sel]
    shifted_pts = np.concatenate([shifted_pts, shifted_pts * np.array([[-1, 1, 1]])], axis=0)
    attn = np.tile(attn, (2, 1))
    bandwidth = estimate_bandwidth(shifted_pts, bandwidth_quantile)
    modes = meanshift_cluster(shifted_pts, bandwidth, attn, max_iter=max_iter)
    joints, _, _ = nms_meanshift(modes, attn, bandwidth, threshold2)
    joints, _ = flip(joints)
    return dict(bandwidth=bandwidth, modes=modes, attn=attn, joints=joints)
