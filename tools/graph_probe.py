"""Does replaying one forward as a hipGraph beat the eager launch sequence? (inter-kernel gaps: DESIGN.md section 5 (10))"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from morig_amd import models, native, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
data = bench.build_batch([1000 + i for i in range(B)], 64).to(dev)
model = synth.load_recipe(models.jointnet_motion(num_keyframes=5, chn_output=3, aggr_method="attn").eval(), 0, mild=True).to(dev)
ops = native.get_ops()

def fwd():
    return model._forward(data, data.pred_flow)          # the plan without the guard's host read

def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

with torch.no_grad():
    for _ in range(3):
        fwd()
    torch.cuda.synchronize()
    print("eager (no guard read)  %.3f ms" % timeit(fwd))
    print("eager module forward   %.3f ms" % timeit(lambda: model(data, data.pred_flow)))
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            fwd()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = fwd()
    torch.cuda.synchronize()
    print("graph replay           %.3f ms" % timeit(g.replay))
    ref = model(data, data.pred_flow)
    g.replay(); torch.cuda.synchronize()
    print("graph == eager:", all(torch.equal(a, b) for a, b in zip(out, ref)))
