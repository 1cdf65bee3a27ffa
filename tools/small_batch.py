"""The one-mesh-per-GPU regime of north_star: jointnet forward at B = 1, 2, 4, 8, 16, 64 meshes per forward on one
MI355X: ms per forward, meshes/s, host time to enqueue one forward, and the per-kind kernel breakdown at the small sizes.
Usage (gpurun): python tools/small_batch.py [B ...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from morig_amd import models, native, synth  # noqa: E402

with_graph = "graph" in sys.argv[1:]
sizes = [int(a) for a in sys.argv[1:] if a != "graph"] or [1, 2, 4, 8, 16, 64]
dev = torch.device("cuda:0")
m = models.jointnet_motion(num_keyframes=5, chn_output=3, aggr_method="attn").eval()
synth.load_recipe(m, 0, mild=True).to(dev)
ops = native.get_ops()
base = None
with torch.no_grad():
    for nb in sizes:
        d = synth.make_batch(range(1000, 1000 + nb), n_side=64, with_skin=False).to(dev)
        d.num_graphs = nb
        for _ in range(5):
            m(d, d.pred_flow)
        torch.cuda.synchronize()
        iters = 40 if nb <= 16 else 15
        t0 = time.perf_counter()
        for _ in range(iters):
            m(d, d.pred_flow)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / iters * 1e3
        native.prof_reset(); native.prof_enable(True)
        for _ in range(3):
            m(d, d.pred_flow)
        torch.cuda.synchronize()
        native.prof_enable(False)
        prof = native.prof_collect()
        kern = sum(v["ms"] for v in prof.values()) / 3
        nl = sum(v["launches"] for v in prof.values()) / 3
        per_mesh = ms / nb
        print(f"B={nb:3d}  {ms:8.3f} ms/forward  {nb / ms * 1e3:8.1f} meshes/s  {per_mesh:6.3f} ms/mesh   sum of kernels {kern:7.3f} ms over {nl:.0f} launches",
              flush=True)
        # the same forwards with the guard read deferred by one forward, and as replays of ONE captured HIP graph
        pend = []
        for _ in range(3):
            m.forward_async(d, d.pred_flow)[1].result()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            pend.append(m.forward_async(d, d.pred_flow)[1])
            if len(pend) > 1:
                assert pend.pop(0).result()
        assert pend.pop(0).result()
        torch.cuda.synchronize()
        ms_def = (time.perf_counter() - t0) / iters * 1e3
        line = f"       deferred guard {ms_def:8.3f} ms/forward ({nb / ms_def * 1e3:7.1f} meshes/s)"
        if with_graph:
            from morig_amd.serving import CapturedForward
            cf = CapturedForward(m, d, d.pred_flow)
            for _ in range(3):
                cf.replay(); cf.check()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                cf.replay()
                assert cf.check()                      # read after every replay (latency of ONE forward, guard included)
            torch.cuda.synchronize()
            ms_g = (time.perf_counter() - t0) / iters * 1e3
            t0 = time.perf_counter()
            for _ in range(iters):
                cf.replay()                            # back to back: the snapshot of the last one is checked
            assert cf.check()
            torch.cuda.synchronize()
            ms_gb = (time.perf_counter() - t0) / iters * 1e3
            line += f" | hipGraph replay {ms_g:8.3f} ms (guard read each) {ms_gb:8.3f} ms back to back ({nb / ms_gb * 1e3:7.1f} meshes/s)"
            del cf
        print(line, flush=True)
        if nb <= 8:
            top = sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:10]
            print("       " + " | ".join(f"{k} {v['ms'] / 3:.3f}" for k, v in top), flush=True)
