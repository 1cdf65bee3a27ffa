"""Multi-GPU sharding for the eval-mode forward: one process per GPU, whole meshes per rank, no
collective inside the network (a mesh's outputs depend only on its own vertices in eval mode,
SURVEY.md 8(e)); ONE all-gather of the per-mesh outputs over RCCL/xGMI at the end.

`torch.distributed` backend "nccl" is RCCL on ROCm; on CPU test runs the same code runs on "gloo".
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.distributed as dist


def shard_items(items: Sequence, rank: int, world: int) -> List:
    """round-robin whole-mesh partition: rank r takes items r, r+world, ..."""
    return [it for i, it in enumerate(items) if i % world == rank]


def unshard_order(n_items: int, world: int) -> List[int]:
    """position in the rank-concatenated order -> original item index."""
    order = []
    for r in range(world):
        order += list(range(r, n_items, world))
    return order


def all_gather_rows(t: torch.Tensor, equal_rows: bool = False, even_alone: bool = False) -> torch.Tensor:
    """Concatenate every rank's [rows_r, C] tensor along dim 0 on every rank.

    equal_rows=True (synthetic batches: same vertex count per rank) is a single
    all_gather_into_tensor; otherwise a count exchange first, then a padded gather (meshes of real
    datasets differ in size). ``even_alone``: issue the collectives at world size 1 too (hardware check of the RCCL
    calls on a one-GPU box)."""
    if not dist.is_available() or not dist.is_initialized() or (dist.get_world_size() == 1 and not even_alone):
        return t
    world = dist.get_world_size()
    t = t.contiguous()
    if equal_rows:
        out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t)
        return out
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    mx = max(counts)
    pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    out = torch.empty((world * mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, pad)
    return torch.cat([out[r * mx: r * mx + c] for r, c in enumerate(counts)], dim=0)
