"""Is a JointNetMotion training step host-bound? Host time until backward() returns against the time until the GPU has drained, and the
back-to-back step time (8 meshes x 4096 vertices). usage: python tools/train_host_time.py  (through gpurun)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morig_amd import models, synth
nb = 8; dev = "cuda"
d = synth.make_batch(range(nb), n_side=64, with_skin=False).to(dev)
m = models.jointnet_motion(num_keyframes=5, chn_output=3, aggr_method="attn").train()
synth.load_recipe(m, 0, mild=True).to(dev)
def step():
    for p in m.parameters(): p.grad = None
    o = m(d, d.pred_flow)
    loss = (o[2] ** 2).mean() + (o[1] ** 2).mean()
    loss.backward()
with torch.enable_grad():
    step(); step(); torch.cuda.synchronize()
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"host enqueue (returns from backward) {1e3*(t1-t0):.1f} ms; GPU drained {1e3*(t2-t0):.1f} ms")
    t0 = time.perf_counter()
    for _ in range(3): step()
    torch.cuda.synchronize(); print(f"3 steps back to back: {1e3*(time.perf_counter()-t0)/3:.1f} ms per step")
