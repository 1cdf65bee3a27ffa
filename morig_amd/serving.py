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
            try:
                with torch.cuda.stream(side):
                    run = lambda: ops.guarded_async(dev, lambda: model._forward(*args, **kwargs))
                    run()                                   # once on the capture stream before recording
                    torch.cuda.synchronize(dev)
                    with torch.cuda.graph(self.graph, stream=side):
                        self.outputs, self.pending = run()
            finally:
                _bm._ctx.key, _bm._ctx.root = outer, None
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
