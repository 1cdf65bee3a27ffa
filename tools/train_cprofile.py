"""cProfile of the HOST side of one JointNetMotion training step (8 meshes), backward on the calling thread. usage: through gpurun"""
import os, sys, time, cProfile, pstats, io
import torch
torch.autograd.set_multithreading_enabled(False)      # the backward runs on this thread: cProfile sees it
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
    pr = cProfile.Profile(); pr.enable(); step(); pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(40); print(s.getvalue()[:8000])
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(30); print(s.getvalue()[:6000])
