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


class RcclComm:
    """The same all-gather through the C-ABI's own RCCL wrapper (include/morig_hip.h: morig_rccl_* / morig_allgather_rows), for a
    host that owns its communicator instead of a torch.distributed process group. ``unique_id`` (128 bytes from
    ``RcclComm.unique_id()`` on rank 0) reaches the other ranks by the host's own means -- here, when torch.distributed is
    initialised, by one broadcast. One communicator per process; calls enqueue on torch's current stream."""

    def __init__(self, n_ranks: int, rank: int, unique_id: bytes):
        import ctypes as C
        from . import native
        self._lib = native.load_library()
        self._native = native
        self.n_ranks, self.rank = n_ranks, rank
        comm = C.c_void_p()
        buf = C.create_string_buffer(bytes(unique_id), 128)
        native.check(self._lib.morig_rccl_comm_init(n_ranks, rank, buf, C.byref(comm)), "morig_rccl_comm_init")
        self._comm = comm

    @staticmethod
    def unique_id() -> bytes:
        import ctypes as C
        from . import native
        buf = C.create_string_buffer(128)
        native.check(native.load_library().morig_rccl_unique_id(buf), "morig_rccl_unique_id")
        return buf.raw

    @classmethod
    def from_process_group(cls):
        """one communicator per rank of the initialised torch.distributed group (the id travels by its broadcast)"""
        rank, world = dist.get_rank(), dist.get_world_size()
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return cls(world, rank, box[0])

    def all_gather_rows(self, t: torch.Tensor) -> torch.Tensor:
        """equal row counts per rank: [rows, C] float32 -> [n_ranks * rows, C], rank-major"""
        import ctypes as C
        assert t.is_cuda and t.dtype == torch.float32 and t.dim() == 2
        t = t.contiguous()
        out = torch.empty((self.n_ranks * t.shape[0], t.shape[1]), dtype=t.dtype, device=t.device)
        self._native.check(self._lib.morig_allgather_rows(self._comm, C.c_void_p(t.data_ptr()), C.c_void_p(out.data_ptr()), t.shape[0],
                                                          t.shape[1], C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                           "morig_allgather_rows")
        return out

    def close(self):
        if self._comm is not None and self._comm.value:
            self._lib.morig_rccl_comm_destroy(self._comm)
            self._comm = None
