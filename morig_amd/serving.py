"""Serving-loop helpers for the eval forwards (north_star's one-mesh-per-GPU operating point, where a forward is 74 launches
of a few tens of microseconds each and the host read of the guard at its end is 10 % of it).

``CapturedForward``: one eval forward on inputs of FIXED sizes captured into a HIP graph (``torch.cuda.CUDAGraph`` = hipGraph):
``replay()`` re-runs all of it -- COO -> CSR builds, every kernel, the guard snapshot -- with one host call. The inputs are the
tensors the capture saw: refresh them in place (``data.pos.copy_(...)``) between replays. Replays are bit-identical to the eager
forward (tests/test_gpu_networks.py::test_captured_forward_replays_bit_identically): the only replay-to-replay differences inside
are the order of a target's edges within its CSR segment (the fill pass claims slots with an atomic cursor; max-aggregation does
not see the order) and never-read padding of ``torch.empty`` buffers.
"""
from __future__ import annotations

import torch

from .models import basic_modules as _bm
from .runtime import get_ops


class CapturedForward:
    def __init__(self, model, *args, warmup: int = 2, **kwargs):
        assert not model.training, "eval-mode forwards only"
        self.model, self.args, self.kwargs = model, args, kwargs
        ops = get_ops()
        dev = next(model.parameters()).device
        self.device = dev
        with torch.no_grad(), torch.cuda.device(dev):
            for _ in range(max(1, warmup)):                 # packs the weights, sizes the allocator's pools
                model(*args, **kwargs)
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            self.graph = torch.cuda.CUDAGraph()
            outer = _bm._ctx.key
            _bm._ctx.key, _bm._ctx.root = model._param_key(), model     # the packed weights are current: no walk inside the capture
            # the range shift the model settled on (NativeModule.forward): the capture runs the stacks at the same 2^-k
            self.range_shift = model.range_shift if getattr(model, "_RANGE_SHIFT_ROOT", False) else 0
            _bm._ctx.shift = self.range_shift
            try:
                with torch.cuda.stream(side):
                    run = lambda: ops.guarded_async(dev, lambda: model._forward(*args, **kwargs))
                    run()                                   # once on the capture stream before recording
                    torch.cuda.synchronize(dev)
                    with torch.cuda.graph(self.graph, stream=side):
                        self.outputs, self.pending = run()
            finally:
                _bm._ctx.key, _bm._ctx.root = outer, None
                _bm._ctx.shift = 0
            torch.cuda.current_stream(dev).wait_stream(side)
        self._key = model._param_key()

    def replay(self):
        """enqueue one forward; -> the (static) output tensors. ``check()`` reads its guard."""
        k = self.model._param_key()
        if k is not self._key and k != self._key:
            raise RuntimeError("the model's parameters changed since the capture: capture again")
        self.pending._done = None
        self.graph.replay()
        return self.outputs

    def check(self) -> bool:
        """the deferred guard read of the last replay (see morig_amd.native.PendingGuard.result)"""
        return self.pending.result()


_DATA_ATTRS = ("pos", "batch", "tpl_edge_index", "geo_edge_index", "skin_input")


class _StaticData:
    """attribute bag with the tensors a rig network's forward reads (morig_amd.synth.MeshData / PyG Batch stand-in)"""


class ForwardServer:
    """``server(data, input_flow) -> outputs`` for a stream of SMALL batches of the rig networks (jointnet / masknet / skinnet): the
    one-mesh-per-GPU operating point north_star describes, where an eager forward is ~110 launches of a few microseconds each and the
    launch path, not the GPU, sets the latency.

    Batches of at most ``graph_max_vertices`` vertices (default 8 x 4096: up to eight 4 k-vertex meshes) are served from a captured HIP
    graph: the first batch of a SHAPE -- (vertices, tpl edges, geo edges, meshes, flow columns) -- is captured into static input buffers
    (``CapturedForward``), every later batch of that shape is copied into them (five small device copies) and replayed with one host
    call; the guard word is read after the replay (an operand outside the split-fp16 range: the batch is re-run eagerly, which falls
    back to the exact-fp32 kernels). At most ``max_graphs`` shapes are kept (least recently used first out). Larger batches -- and
    every batch while ``enabled`` is False -- run the eager forward: there the GPU time dominates and a graph buys nothing (measured:
    B = 8 1 % , B = 1 10-12 %).

    The returned tensors of a served batch are the graph's STATIC outputs: they are overwritten by the next call with the same shape
    (``copy_outputs=True`` returns clones instead)."""

    def __init__(self, model, graph_max_vertices: int = 8 * 4096, max_graphs: int = 8, copy_outputs: bool = False):
        assert not model.training, "eval-mode forwards only"
        self.model, self.graph_max_vertices, self.max_graphs, self.copy_outputs = model, graph_max_vertices, max_graphs, copy_outputs
        self.enabled = True
        self._graphs = {}                                   # shape key -> (CapturedForward, static data, static flow), LRU order
        self.stats = dict(captures=0, replays=0, eager=0, fallbacks=0)

    @staticmethod
    def _key(data, flow):
        ng = getattr(data, "num_graphs", None)
        if ng is None:
            ng = int(data.batch.max().item()) + 1          # one host read per batch, as the eager forward does without num_graphs
        skin = getattr(data, "skin_input", None)
        return (int(data.pos.shape[0]), int(data.tpl_edge_index.shape[1]), int(data.geo_edge_index.shape[1]), int(ng),
                tuple(flow.shape), None if skin is None else tuple(skin.shape))

    def _capture(self, key, data, flow):
        sd = _StaticData()
        for k in _DATA_ATTRS:
            t = getattr(data, k, None)
            if torch.is_tensor(t):
                setattr(sd, k, t.clone())
        sd.num_graphs = key[3]
        sflow = flow.clone()
        cf = CapturedForward(self.model, sd, sflow)
        self.stats["captures"] += 1
        while len(self._graphs) >= self.max_graphs:
            self._graphs.pop(next(iter(self._graphs)))
        self._graphs[key] = (cf, sd, sflow)
        return self._graphs[key]

    def __call__(self, data, input_flow):
        n = int(data.pos.shape[0])
        if not self.enabled or n > self.graph_max_vertices:
            self.stats["eager"] += 1
            with torch.no_grad():
                return self.model(data, input_flow)
        key = self._key(data, input_flow)
        hit = self._graphs.pop(key, None)
        if hit is None:
            try:
                hit = self._capture(key, data, input_flow)
            except RuntimeError:                            # parameters changed under an old capture, or the capture failed: serve eagerly
                self.stats["eager"] += 1
                with torch.no_grad():
                    return self.model(data, input_flow)
        else:
            self._graphs[key] = hit                         # most recently used last
            cf, sd, sflow = hit
            for k in _DATA_ATTRS:
                t = getattr(sd, k, None)
                if t is not None:
                    t.copy_(getattr(data, k), non_blocking=True)
            sflow.copy_(input_flow, non_blocking=True)
        cf = hit[0]
        try:
            out = cf.replay()
        except RuntimeError:                                # the model's parameters changed since the capture: drop it, capture again
            self._graphs.pop(key, None)
            return self(data, input_flow)
        self.stats["replays"] += 1
        if not cf.check():
            # an operand left the split-fp16 range: the eager forward answers (it raises the model's range shift, or re-runs on the exact
            # path); the graph was captured at the old shift and would overflow again -- dropped, the next batch of this shape captures anew
            self.stats["fallbacks"] += 1
            self._graphs.pop(key, None)
            with torch.no_grad():
                return self.model(data, input_flow)
        return tuple(o.clone() for o in out) if self.copy_outputs else out
