import os, sys, time, cProfile, pstats, io
import torch
sys.path.insert(0, "/root/repo")
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
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45); print(s.getvalue()[:9000])
