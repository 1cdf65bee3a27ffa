"""Device-side producer of ``geo_edge_index``: the reference's ``get_geo_edges`` (/root/reference/data_proc/common_ops.py:214-226),
batched over the meshes of a batch and run by ``morig_geo_ball_graph`` (csrc/graph.hip).

    get_geo_edges(pos, batch, radius=0.06, max_nn=15)             Euclidean balls (SURVEY 8(d)'s synthetic recipe)
    get_geo_edges_from_distance(dist, radius=0.06, max_nn=15)     the reference's own input: an n x n geodesic distance matrix

Both return an int64 ``[2, E]`` device tensor: row 0 = vertex i, row 1 = ball member -- the layout ``np.loadtxt(geo_e).T`` has in
``datasets/dataset_rig.py:86`` (the reference function itself returns the transpose, ``[E, 2]`` rows ``[i, member]``).
``self_loops=True`` appends the pairs (i, i) that ``add_self_loops`` appends in ``dataset_rig.py:122``.
Rows with more than ``max_nn`` members keep a uniformly random subset: the reference draws it from numpy's global stream, this
kernel from a counter hash of (seed, row, member number); ``seed=None`` takes one draw from torch's generator, so
``torch.manual_seed`` makes the graph reproducible.
"""
from __future__ import annotations

from typing import Optional

import torch

from .native import Mat
from .runtime import get_ops


def _seed(seed: Optional[int]) -> int:
    return int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if seed is None else int(seed)


def mesh_ptr_of(batch: Optional[torch.Tensor], n: int, device, num_graphs: Optional[int] = None) -> torch.Tensor:
    """int32 [B + 1] row offsets of PyG's sorted ``batch`` vector (None: one mesh)"""
    if batch is None:
        return torch.tensor([0, n], dtype=torch.int32, device=device)
    counts = torch.bincount(batch, minlength=num_graphs or 0)
    ptr = torch.zeros(counts.numel() + 1, dtype=torch.int32, device=device)
    ptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
    return ptr


def get_geo_edges(pos: torch.Tensor, batch: Optional[torch.Tensor] = None, radius: float = 0.06, max_nn: int = 15,
                  seed: Optional[int] = None, self_loops: bool = False, num_graphs: Optional[int] = None,
                  return_members: bool = False):
    ops = get_ops()
    p = pos.float().contiguous()
    ptr = mesh_ptr_of(batch, p.shape[0], p.device, num_graphs)
    ei, members = ops.geo_ball_graph(Mat.of(p, 0, 3), ptr, radius, max_nn, _seed(seed), self_loops=self_loops)
    return (ei, members) if return_members else ei


def get_geo_edges_from_distance(dist: torch.Tensor, radius: float = 0.06, max_nn: int = 15, seed: Optional[int] = None,
                                self_loops: bool = False, return_members: bool = False):
    ops = get_ops()
    ei, members = ops.geo_ball_graph(None, None, radius, max_nn, _seed(seed), self_loops=self_loops,
                                     dist=dist.double().contiguous())
    return (ei, members) if return_members else ei
