"""Seeded synthetic inputs for the MoRig geometric-network forward path.

Everything here is *harness* code (CPU, numpy/torch): the synthetic mesh recipe of
SURVEY.md section 8(d), PyG-style batch collation, and a construction-order-independent
weight/BatchNorm-statistics recipe so that the oracle, the golden-vector generator, the
tests and bench.py all see bit-identical parameters without shipping 30 MB state_dicts.

Reference semantics mirrored (cited, not copied):
  * tpl edges  : rows ``[v, n]`` for every 1-ring neighbour      data_proc/common_ops.py:15-32
  * geo edges  : rows ``[i, j]`` for <=15 random members of the
                 radius-0.06 ball around i (self excluded)        data_proc/common_ops.py:214-226
  * datasets add one self loop per node before batching           datasets/dataset_rig.py:121-122
  * PyG collation: node tensors concatenated on dim 0, ``*_index``
    tensors concatenated on dim -1 and offset by the node count   (PyG Batch)
"""
from __future__ import annotations

import math
import zlib
from types import SimpleNamespace
from typing import Dict, Iterable, List, Optional

import numpy as np
import torch

__all__ = [
    "MeshData", "make_mesh", "make_point_cloud", "collate", "make_batch", "make_batch_device",
    "recipe_state_dict", "load_recipe", "KEYFRAMES",
]

KEYFRAMES = 5


class MeshData(SimpleNamespace):
    """Duck-typed stand-in for ``torch_geometric.data.Data`` / ``Batch``.

    The reference only ever reads attributes (``data.pos``, ``data.batch`` ...) and calls
    ``data.to(device)`` (training/train_rig.py:207), so that is all this provides."""

    def to(self, device):
        out = MeshData()
        for k, v in self.__dict__.items():
            setattr(out, k, v.to(device) if torch.is_tensor(v) else v)
        return out

    def keys(self):
        return list(self.__dict__.keys())


def _torus(n_side: int, R: float, r: float) -> np.ndarray:
    u = (np.arange(n_side) / n_side) * 2.0 * np.pi
    uu, vv = np.meshgrid(u, u, indexing="ij")
    x = (R + r * np.cos(vv)) * np.cos(uu)
    z = (R + r * np.cos(vv)) * np.sin(uu)
    y = r * np.sin(vv) + r          # lifted so that y >= 0 (normalize(): common_ops.py:123-138)
    return np.stack([x, y, z], axis=-1).reshape(-1, 3)


def make_mesh(seed: int, n_side: int = 64, geo_radius: Optional[float] = None,
              geo_max_nn: int = 15, with_skin: bool = True, geo: str = "host") -> MeshData:
    """One synthetic character-like mesh: an ``n_side x n_side`` triangulated torus grid.
    geo="host": the ball graph by a V x V distance matrix on the CPU (tests, oracle, CPU baseline); geo="none": only the self
    loops -- ``make_batch_device`` then builds the ball graph of the whole batch on the GPU (morig_amd/graph_build.py)."""
    rng = np.random.default_rng([0x4D6F5269, seed])
    R = 0.35 * (1.0 + 0.1 * rng.uniform(-1, 1))
    r = 0.12 * (1.0 + 0.1 * rng.uniform(-1, 1))
    pos = _torus(n_side, R, r) + rng.normal(0.0, 1e-3, size=(n_side * n_side, 3))
    pos = pos.astype(np.float32)
    V = pos.shape[0]

    # 1-ring of a regular triangulated grid: 6 neighbours, rows [v, n]
    idx = np.arange(V).reshape(n_side, n_side)
    nbrs = []
    for du, dv in ((1, 0), (-1, 0), (0, 1), (0, -1), (1, 1), (-1, -1)):
        nbrs.append(np.roll(np.roll(idx, -du, axis=0), -dv, axis=1).reshape(-1))
    tpl = np.stack([np.repeat(np.arange(V), 6), np.stack(nbrs, axis=1).reshape(-1)], axis=0)

    # geodesic-ball edges, Euclidean distance standing in for geodesic distance
    if geo_radius is None:
        geo_radius = 0.06 * 64.0 / n_side
    if geo == "host":
        pt = torch.from_numpy(pos)
        d2 = torch.cdist(pt, pt, compute_mode="donot_use_mm_for_euclid_dist") ** 2
        d2.fill_diagonal_(1e9)
        # random subset of <= geo_max_nn ball members per row (np.random.choice in the reference):
        # the members with the smallest random keys
        keys = torch.from_numpy(rng.random((V, V), dtype=np.float32))
        keys[d2 > geo_radius * geo_radius] = 2.0
        k = min(geo_max_nn, V - 1)
        val, pick = torch.topk(keys, k, dim=1, largest=False, sorted=True)
        valid = val < 2.0
        rows = torch.arange(V)[:, None].expand(-1, k)
        geo = torch.stack([rows[valid], pick[valid]], dim=0).numpy().astype(np.int64)
    else:
        assert geo == "none"
        geo = np.zeros((2, 0), dtype=np.int64)

    loops = np.stack([np.arange(V), np.arange(V)], axis=0)
    tpl = np.concatenate([tpl, loops], axis=1)     # datasets/dataset_rig.py:121
    geo = np.concatenate([geo, loops], axis=1)     # datasets/dataset_rig.py:122

    d = MeshData(
        pos=torch.from_numpy(pos),
        tpl_edge_index=torch.from_numpy(tpl).long(),
        geo_edge_index=torch.from_numpy(geo).long(),
        pred_flow=torch.from_numpy(rng.normal(0.0, 0.05, size=(V, 3 * KEYFRAMES)).astype(np.float32)),
        name=seed,
    )
    if with_skin:
        # 20 nearest bones x (6 coords, 1/D_g, leaf flag)     datasets/dataset_rig.py:31-76
        sk = np.empty((V, 20, 8), dtype=np.float32)
        sk[:, :, 0:6] = rng.uniform(-0.5, 0.5, size=(V, 20, 6))
        sk[:, :, 6] = rng.uniform(1.0, 50.0, size=(V, 20))
        sk[:, :, 7] = rng.integers(0, 2, size=(V, 20))
        d.skin_input = torch.from_numpy(sk.reshape(V, 160))
    return d


def make_point_cloud(mesh: MeshData, seed: int, n_pts: int = 8192, sigma: float = 0.004) -> torch.Tensor:
    """Partial-scan stand-in for CorrNet: random mesh vertices + N(0, sigma^2) noise."""
    rng = np.random.default_rng([0x50747321, seed])
    V = mesh.pos.shape[0]
    pick = rng.integers(0, V, size=n_pts)
    pts = mesh.pos.numpy()[pick] + rng.normal(0.0, sigma, size=(n_pts, 3))
    return torch.from_numpy(pts.astype(np.float32))


def collate(meshes: List[MeshData], clouds: Optional[List[torch.Tensor]] = None) -> MeshData:
    """PyG ``Batch.from_data_list`` semantics for the attributes the hot path reads."""
    out = MeshData()
    off = 0
    pos, tpl, geo, flow, batch, skin = [], [], [], [], [], []
    for b, m in enumerate(meshes):
        V = m.pos.shape[0]
        pos.append(m.pos)
        tpl.append(m.tpl_edge_index + off)
        geo.append(m.geo_edge_index + off)
        flow.append(m.pred_flow)
        batch.append(torch.full((V,), b, dtype=torch.long))
        if hasattr(m, "skin_input"):
            skin.append(m.skin_input)
        off += V
    out.pos = torch.cat(pos, 0)
    out.tpl_edge_index = torch.cat(tpl, 1)
    out.geo_edge_index = torch.cat(geo, 1)
    out.pred_flow = torch.cat(flow, 0)
    out.batch = torch.cat(batch, 0)
    out.name = torch.tensor([int(m.name) for m in meshes])
    out.num_graphs = len(meshes)                 # as torch_geometric's Batch carries it (saves the dim_size host read)
    if skin:
        out.skin_input = torch.cat(skin, 0)
    # CorrNet naming (datasets/dataset_pose.py: vtx / pts / vtx_batch / pts_batch)
    out.vtx = out.pos
    out.vtx_batch = out.batch
    if clouds is not None:
        out.pts = torch.cat(clouds, 0)
        out.pts_batch = torch.cat([torch.full((c.shape[0],), b, dtype=torch.long)
                                   for b, c in enumerate(clouds)], 0)
    return out


def make_batch(seeds: Iterable[int], n_side: int = 64, n_pts: int = 0, **kw) -> MeshData:
    meshes = [make_mesh(s, n_side=n_side, **kw) for s in seeds]
    clouds = [make_point_cloud(m, int(m.name), n_pts) for m in meshes] if n_pts else None
    return collate(meshes, clouds)


def make_batch_device(seeds: Iterable[int], device, n_side: int = 64, n_pts: int = 0, geo_radius: Optional[float] = None,
                      geo_max_nn: int = 15, geo_seed: int = 0, **kw) -> MeshData:
    """The same synthetic batch with its ``geo_edge_index`` built ON THE DEVICE for all meshes at once
    (``graph_build.get_geo_edges`` = morig_geo_ball_graph: the reference's get_geo_edges, data_proc/common_ops.py:214-226, then
    the datasets' self loops): positions, the 1-ring graph and the flows come from the host recipe, the V x V ball search --
    the only O(V^2) part -- never runs on the CPU. Needs the HIP library and a GPU (no fallback)."""
    from . import graph_build
    meshes = [make_mesh(s, n_side=n_side, geo="none", **kw) for s in seeds]
    clouds = [make_point_cloud(m, int(m.name), n_pts) for m in meshes] if n_pts else None
    b = collate(meshes, clouds).to(device)
    r = 0.06 * 64.0 / n_side if geo_radius is None else geo_radius
    b.geo_edge_index = graph_build.get_geo_edges(b.pos, b.batch, r, geo_max_nn, seed=geo_seed, self_loops=True,
                                                 num_graphs=b.num_graphs)
    return b


# --------------------------------------------------------------------------------------
# parameter recipe
# --------------------------------------------------------------------------------------

def _rng_for(seed: int, key: str) -> np.random.Generator:
    return np.random.default_rng([seed, zlib.crc32(key.encode("utf-8"))])


def recipe_state_dict(template: Dict[str, torch.Tensor], seed: int, mild: bool = False) -> Dict[str, torch.Tensor]:
    """Values for every entry of ``template`` (a ``state_dict()``) as a pure function of
    (seed, key, shape): independent of module construction order and of torch's RNG.

    BatchNorm entries get *non-trivial* statistics (default-initialised BN
    is an identity-like affine, which hides folding bugs): mean ~ N(0, .3), var ~ U(.3, 2),
    gamma ~ N(0, 1) clipped away from 0 with both signs (``mild``: U(.5, 1.5)), beta ~ N(0, .2).
    """
    bn_prefixes = {k[: -len(".running_mean")] for k in template if k.endswith(".running_mean")}
    out: Dict[str, torch.Tensor] = {}
    for key, ref in template.items():
        rng = _rng_for(seed, key)
        shape = tuple(ref.shape)
        prefix, _, leaf = key.rpartition(".")
        if leaf == "num_batches_tracked":
            out[key] = torch.zeros_like(ref)
            continue
        if prefix in bn_prefixes:
            if leaf == "running_mean":
                v = rng.normal(0.0, 0.3, size=shape)
            elif leaf == "running_var":
                v = rng.uniform(0.3, 2.0, size=shape)
            elif leaf == "weight":
                if mild:
                    v = rng.uniform(0.5, 1.5, size=shape)
                else:
                    v = rng.normal(0.0, 1.0, size=shape)
                    v = np.sign(v) * np.maximum(np.abs(v), 0.05)
            else:  # bias
                v = rng.normal(0.0, 0.2, size=shape)
        elif key.endswith("cls_token"):
            v = rng.normal(0.0, 1.0, size=shape)
        elif key.endswith("temprature"):
            out[key] = ref.clone()
            continue
        elif len(shape) == 2:                       # Linear.weight  [out, in]
            bound = math.sqrt(6.0 / shape[1])       # He-uniform keeps post-ReLU scale O(1)
            v = rng.uniform(-bound, bound, size=shape)
        else:                                       # Linear.bias: fan_in unknown here -> small
            v = rng.uniform(-0.1, 0.1, size=shape)
        out[key] = torch.from_numpy(np.asarray(v, dtype=np.float32)).reshape(shape).clone()
    return out


def load_recipe(model: torch.nn.Module, seed: int, mild: bool = False) -> torch.nn.Module:
    sd = recipe_state_dict(model.state_dict(), seed, mild=mild)
    missing = model.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return model
