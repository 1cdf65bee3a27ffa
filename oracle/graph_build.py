"""TEST INFRASTRUCTURE ONLY -- CPU restatement (numpy) of the geodesic-ball graph build that produces `geo_edge_index`.

Follows /root/reference/data_proc/common_ops.py:214-226 (`get_geo_edges`): the distance matrix gets +10 on its diagonal (:218),
row i keeps the indices with dist <= radius in index order (:220), a row with more than max_nn members keeps
`np.random.choice(members, max_nn, replace=False)` (:221-222, numpy's global stream), rows are `[i, member]` (:223-224).
Pinned against the reference's own function by tests/golden/geo_edges_kat.npz (oracle/make_golden.py geo_edges).

`euclid_sq_f32` is the distance the device kernel's positions variant uses in the geodesic's place (SURVEY 8(d) synthetic
recipe): d^2 = (dx*dx + dy*dy) + dz*dz in float32, compared against float32(radius)^2.
"""
from __future__ import annotations

import numpy as np


def get_geo_edges_from_distance(dist: np.ndarray, radius: float = 0.06, max_nn: int = 15, rng=None) -> np.ndarray:
    """-> int64 [E, 2] rows [i, member]; rng: object with .choice (default: numpy's global stream, as the reference)"""
    d = np.array(dist, dtype=np.float64, copy=True)
    d += 10.0 * np.eye(len(d))                                   # :218 remove self-loop edge
    choose = (rng or np.random).choice
    rows = []
    for i in range(len(d)):
        members = np.argwhere(d[i, :] <= radius).squeeze(1)      # :220
        if len(members) > max_nn:
            members = choose(members, max_nn, replace=False)     # :221-222
        rows.append(np.stack([np.full(len(members), i, dtype=np.int64), members.astype(np.int64)], axis=1))
    return np.concatenate(rows, axis=0) if rows else np.zeros((0, 2), np.int64)


def euclid_sq_f32(pos: np.ndarray) -> np.ndarray:
    """[n, n] float32 squared distances, each evaluated as (dx*dx + dy*dy) + dz*dz with float32 roundings (no fma)"""
    p = np.asarray(pos, dtype=np.float32)
    dx = p[:, None, 0] - p[None, :, 0]
    dy = p[:, None, 1] - p[None, :, 1]
    dz = p[:, None, 2] - p[None, :, 2]
    return (dx * dx + dy * dy) + dz * dz


def member_lists(pos: np.ndarray, mesh_ptr, radius: float):
    """per vertex the sorted member indices (global row ids) of its ball inside its own mesh, positions variant"""
    r2 = np.float32(radius) * np.float32(radius)
    out = []
    for b in range(len(mesh_ptr) - 1):
        s, e = int(mesh_ptr[b]), int(mesh_ptr[b + 1])
        d2 = euclid_sq_f32(pos[s:e])
        hit = d2 <= r2
        np.fill_diagonal(hit, False)
        out.extend([s + np.flatnonzero(hit[i]) for i in range(e - s)])
    return out
