"""TEST INFRASTRUCTURE ONLY -- never imported by the product (``morig_amd``).

Pure-torch CPU restatements of the *third-party* operators the MoRig hot path calls. Their
sources are NOT under /root/reference (un-vendored wheels pinned in environment.yml:91,100-105:
pyg 2.0.4, pytorch-scatter 2.0.9, pytorch-cluster 1.6.0), so each function restates the published
algorithm of that pinned version and names the reference call site it serves.

PARITY STATUS: the reference holds no tests, golden vectors or fixtures for these call sites
(SURVEY.md section 4), and the wheels cannot be installed here (no network). These restatements
are therefore pinned only by the hand-computed known-answer tests in tests/test_oracle_kat.py.
"parity unpinned" applies to exactly this file; everything *above* it (layer wiring, slicing,
concat order, attention, normalisation) is pinned by running the reference's own
models/*.py unmodified on top of these functions (oracle/shim.py, oracle/make_golden.py).
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch


# ----------------------------------------------------------------------------- utils
def remove_self_loops(edge_index: torch.Tensor, edge_attr=None):
    """torch_geometric.utils.remove_self_loops -- call sites models/basic_modules.py:149,188."""
    keep = edge_index[0] != edge_index[1]
    return edge_index[:, keep], (None if edge_attr is None else edge_attr[keep])


def add_self_loops(edge_index: torch.Tensor, edge_attr=None, fill_value=None, num_nodes: Optional[int] = None):
    """torch_geometric.utils.add_self_loops -- call sites models/basic_modules.py:150,189,
    datasets/dataset_rig.py:121-122. Appends (i, i) for i in [0, num_nodes)."""
    if num_nodes is None:
        num_nodes = int(edge_index.max()) + 1 if edge_index.numel() else 0
    loop = torch.arange(num_nodes, dtype=edge_index.dtype, device=edge_index.device)
    return torch.cat([edge_index, torch.stack([loop, loop], 0)], dim=1), edge_attr


# ----------------------------------------------------------------------------- scatter
def scatter_max(src: torch.Tensor, index: torch.Tensor, dim: int = 0, dim_size: Optional[int] = None):
    """torch_scatter.scatter_max(src, index, dim=0) -- call sites models/rignet.py:63,176,
    models/corrnet.py:44. Returns (values, argmax-placeholder). Empty segments -> 0."""
    assert dim == 0
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() else 0
    out = src.new_full((dim_size,) + tuple(src.shape[1:]), float("-inf"))
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    out = out.scatter_reduce(0, idx, src, reduce="amax", include_self=True)
    out = torch.where(torch.isinf(out) & (out < 0), torch.zeros_like(out), out)
    return out, None


def scatter_add(src: torch.Tensor, index: torch.Tensor, dim: int = 0, dim_size: Optional[int] = None):
    """torch_scatter.scatter_add -- used by knn_interpolate (PyG) and models/deformnet.py:54."""
    assert dim == 0
    if dim_size is None:
        dim_size = int(index.max()) + 1
    out = src.new_zeros((dim_size,) + tuple(src.shape[1:]))
    return out.index_add(0, index, src)


def global_max_pool(x: torch.Tensor, batch: torch.Tensor):
    """torch_geometric.nn.global_max_pool -- call site models/basic_modules.py:122."""
    return scatter_max(x, batch, dim=0, dim_size=int(batch.max()) + 1)[0]


def propagate_max(message: torch.Tensor, target_index: torch.Tensor, num_targets: int) -> torch.Tensor:
    """MessagePassing(aggr='max').aggregate: per-channel max of messages sharing a target;
    targets with no incoming message get 0 (torch_scatter 'max' fill)."""
    return scatter_max(message, target_index, dim=0, dim_size=num_targets)[0]


# ----------------------------------------------------------------------------- cluster
def _segments(batch: torch.Tensor):
    """(start, end) of each batch id; ``batch`` is sorted (PyG collation)."""
    nb = int(batch.max()) + 1 if batch.numel() else 0
    counts = torch.bincount(batch, minlength=nb)
    ends = torch.cumsum(counts, 0)
    starts = ends - counts
    return [(int(s), int(e)) for s, e in zip(starts, ends)]


def fps(pos: torch.Tensor, batch: Optional[torch.Tensor] = None, ratio: float = 0.5, random_start: bool = True):
    """torch_cluster.fps (1.6.0) -- call sites models/basic_modules.py:75,99.
    Per cloud: ceil(ratio * n) samples; start = first point (random_start=False) or a uniformly
    random one; then repeatedly the point with the largest min-squared-distance to the chosen set
    (first index wins ties). Returns global indices, clouds concatenated in order."""
    if batch is None:
        batch = pos.new_zeros(pos.shape[0], dtype=torch.long)
    out = []
    for s, e in _segments(batch):
        n = e - s
        m = int(math.ceil(ratio * n))
        p = pos[s:e]
        cur = int(torch.randint(n, (1,))) if random_start else 0
        dist = torch.full((n,), float("inf"), dtype=pos.dtype)
        sel = []
        for _ in range(m):
            sel.append(cur + s)
            d = ((p - p[cur]) ** 2).sum(-1)
            dist = torch.minimum(dist, d)
            cur = int(torch.argmax(dist))
        out.append(torch.tensor(sel, dtype=torch.long))
    return torch.cat(out) if out else torch.zeros(0, dtype=torch.long)


def radius(x: torch.Tensor, y: torch.Tensor, r: float, batch_x=None, batch_y=None, max_num_neighbors: int = 32):
    """torch_cluster.radius (1.6.0, CUDA kernel semantics -- the path the reference takes on a GPU,
    models/basic_modules.py:77): for each y, scan x of the same cloud in index order and keep the
    first ``max_num_neighbors`` with squared distance < r*r (strict). Returns (row=y_idx, col=x_idx)."""
    if batch_x is None:
        batch_x = x.new_zeros(x.shape[0], dtype=torch.long)
    if batch_y is None:
        batch_y = y.new_zeros(y.shape[0], dtype=torch.long)
    segx = _segments(batch_x)
    rows, cols = [], []
    for b, (ys, ye) in enumerate(_segments(batch_y)):
        if b >= len(segx):
            break
        xs, xe = segx[b]
        d2 = ((y[ys:ye, None, :] - x[None, xs:xe, :]) ** 2).sum(-1)
        hit = d2 < r * r
        rank = torch.cumsum(hit.long(), dim=1)
        keep = hit & (rank <= max_num_neighbors)
        yi, xi = torch.nonzero(keep, as_tuple=True)
        rows.append(yi + ys)
        cols.append(xi + xs)
    return torch.cat(rows), torch.cat(cols)


def knn(x: torch.Tensor, y: torch.Tensor, k: int, batch_x=None, batch_y=None, cosine: bool = False):
    """torch_cluster.knn (1.6.0) -- call sites models/corrnet.py:64 (cosine=True, k=1) and inside
    knn_interpolate. For each y the k nearest x of the same cloud (Euclidean, or 1 - cosine
    similarity); nearest first; lowest index wins ties. Returns [y_idx ; x_idx]."""
    if batch_x is None:
        batch_x = x.new_zeros(x.shape[0], dtype=torch.long)
    if batch_y is None:
        batch_y = y.new_zeros(y.shape[0], dtype=torch.long)
    segx = _segments(batch_x)
    rows, cols = [], []
    for b, (ys, ye) in enumerate(_segments(batch_y)):
        if b >= len(segx):
            break
        xs, xe = segx[b]
        if xe == xs or ye == ys:
            continue
        xa, ya = x[xs:xe], y[ys:ye]
        if cosine:
            xn = xa / xa.norm(dim=1, keepdim=True)
            yn = ya / ya.norm(dim=1, keepdim=True)
            d = 1.0 - yn @ xn.t()
        else:
            d = ((ya[:, None, :] - xa[None, :, :]) ** 2).sum(-1)
        kk = min(k, xe - xs)
        # stable sort => lowest index on ties
        order = torch.sort(d, dim=1, stable=True)[1][:, :kk]
        rows.append((torch.arange(ys, ye)[:, None].expand(-1, kk)).reshape(-1))
        cols.append((order + xs).reshape(-1))
    return torch.stack([torch.cat(rows), torch.cat(cols)], 0)


def knn_interpolate(x, pos_x, pos_y, batch_x=None, batch_y=None, k: int = 3, num_workers: int = 1):
    """torch_geometric.nn.knn_interpolate (2.0.4) -- call site models/basic_modules.py:134:
    inverse-squared-distance weights w = 1/clamp(d^2, min=1e-16), y = sum(w x)/sum(w)."""
    with torch.no_grad():
        y_idx, x_idx = knn(pos_x, pos_y, k, batch_x=batch_x, batch_y=batch_y)
        diff = pos_x[x_idx] - pos_y[y_idx]
        w = 1.0 / torch.clamp((diff * diff).sum(-1, keepdim=True), min=1e-16)
    num = scatter_add(x[x_idx] * w, y_idx, dim=0, dim_size=pos_y.size(0))
    den = scatter_add(w, y_idx, dim=0, dim_size=pos_y.size(0))
    return num / den


# ----------------------------------------------------------------------------- convs
class MessagePassing(torch.nn.Module):
    """Minimal torch_geometric.nn.conv.MessagePassing (2.0.4) for flow='source_to_target':
    ``*_j`` = tensor[edge_index[0]] (source), ``*_i`` = tensor[edge_index[1]] (target);
    aggregate at the target with dim_size = number of target nodes; then ``update``.
    Serves EdgeConv / EdgeConvMotion (models/basic_modules.py:142-202)."""

    def __init__(self, aggr: str = "max", **kwargs):
        super().__init__()
        assert aggr == "max", "the MoRig hot path only uses aggr='max'"
        self.aggr = aggr

    def propagate(self, edge_index, size=None, **kwargs):
        import inspect
        want = list(inspect.signature(self.message).parameters)
        src, dst = edge_index[0], edge_index[1]
        n_dst = None
        args = {}
        for name in want:
            base, side = name[:-2], name[-2:]
            data = kwargs.get(base)
            if isinstance(data, (tuple, list)):
                s_data, d_data = data
                if torch.is_tensor(d_data):
                    n_dst = d_data.size(0)
                pick = s_data if side == "_j" else d_data
            else:
                pick = data
                if torch.is_tensor(data) and n_dst is None:
                    n_dst = data.size(0)
            if pick is None:
                args[name] = None
            else:
                args[name] = pick.index_select(0, src if side == "_j" else dst)
        if n_dst is None:
            raise ValueError("cannot infer number of target nodes")
        msg = self.message(**args)
        out = propagate_max(msg, dst, n_dst)
        return self.update(out)

    def update(self, aggr_out):
        return aggr_out


class PointConv(MessagePassing):
    """torch_geometric.nn.PointConv (2.0.4; later renamed PointNetConv) -- call site
    models/basic_modules.py:72,82-84. Defaults: aggr='max', add_self_loops=True. With a bipartite
    (source, target) input the self-loop step still runs on the raw index pairs: columns with
    edge_index[0]==edge_index[1] are dropped and (k, k) is appended for k < min(N_src, N_dst).
    message = local_nn([x_j ‖ pos_j - pos_i]) (just the offset when x is None)."""

    def __init__(self, local_nn=None, global_nn=None, add_self_loops: bool = True, **kwargs):
        kwargs.setdefault("aggr", "max")
        super().__init__(**kwargs)
        self.local_nn = local_nn
        self.global_nn = global_nn
        self.add_self_loops = add_self_loops

    def forward(self, x, pos, edge_index):
        if not isinstance(x, (tuple, list)):
            x = (x, None)
        if torch.is_tensor(pos):
            pos = (pos, pos)
        if self.add_self_loops:
            edge_index, _ = remove_self_loops(edge_index)
            edge_index, _ = add_self_loops(edge_index, num_nodes=min(pos[0].size(0), pos[1].size(0)))
        out = self.propagate(edge_index, x=tuple(x), pos=tuple(pos))
        if self.global_nn is not None:
            out = self.global_nn(out)
        return out

    def message(self, x_j, pos_i, pos_j):
        msg = pos_j - pos_i
        if x_j is not None:
            msg = torch.cat([x_j, msg], dim=1)
        if self.local_nn is not None:
            msg = self.local_nn(msg)
        return msg


# ---------------------------------------------------------------------------------------------------
# torch_geometric.data stand-ins, just enough for datasets/dataset_rig.py to run its own process()
# ---------------------------------------------------------------------------------------------------
class Data:
    """attribute bag: Data(pos=..., tpl_edge_index=..., ...) (datasets/dataset_rig.py:134-138)."""

    def __init__(self, **kwargs):
        self.__dict__.update(kwargs)

    def keys(self):
        return list(self.__dict__.keys())


class InMemoryDataset:
    """PyG's life cycle as dataset_rig.py relies on it: ``raw_paths`` from ``raw_file_names`` (the reference returns
    absolute glob results), ``processed_paths`` under ``root/processed``, ``process()`` run once when the processed
    file is missing, ``collate`` -> (data, slices). Here collate keeps the per-model list (no concatenation)."""

    def __init__(self, root=None, transform=None, pre_transform=None):
        import os
        self.root = root
        os.makedirs(os.path.join(root, "processed"), exist_ok=True)
        if not os.path.exists(self.processed_paths[0]):
            self.process()

    @property
    def raw_paths(self):
        return sorted(self.raw_file_names)

    @property
    def processed_paths(self):
        import os
        names = self.processed_file_names
        names = [names] if isinstance(names, str) else list(names)
        return [os.path.join(self.root, "processed", n) for n in names]

    @staticmethod
    def collate(data_list):
        return data_list, None
