"""small-batch forwards of jointnet_motion with the two EdgeConvs of a unit on two streams (native.Fork) or on one: eager and
served (captured HIP graph) milliseconds per forward at B = 1, 2, 4, 8, 16 meshes; MORIG_FORK_ROWS is read per process, so this
script is run once per setting (tools/gpu_r06x.sh)"""
import os, sys, time, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from morig_amd import models, synth, native
from morig_amd.serving import ForwardServer
dev = torch.device("cuda", 0)
m = synth.load_recipe(models.jointnet_motion(num_keyframes=5, chn_output=3, aggr_method="attn").eval(), 0, mild=True).to(dev)
res = {"fork_rows": getattr(native, "FORK_ROWS", None)}
ref = {}
with torch.no_grad():
    for nb in [int(a) for a in (sys.argv[1:] or ["1", "2", "4", "8", "16"])]:
        d = bench.build_batch([1000 + i for i in range(nb)], 64, dev=dev)
        def timed(fn, n=40, w=6):
            for _ in range(w): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n): fn()
            torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
        eager = min(timed(lambda: m(d, d.pred_flow)) for _ in range(3))
        srv = ForwardServer(m)
        served = min(timed(lambda: srv(d, d.pred_flow)) for _ in range(3))
        out = srv(d, d.pred_flow)[2].clone()
        res[f"B{nb}"] = dict(eager_ms=round(eager, 3), served_ms=round(served, 3), stats=dict(srv.stats),
                             checksum=float(out.double().abs().sum()))
        del srv
print(json.dumps(res))
