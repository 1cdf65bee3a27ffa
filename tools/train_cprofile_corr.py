"""cProfile of the HOST side of one CorrNet training step (8 pairs: 4 k-vertex mesh + 8 k-point cloud), backward on the calling thread.
usage: python tools/train_cprofile_corr.py  (through gpurun)"""
import os, sys, time, cProfile, pstats, io
import torch
torch.autograd.set_multithreading_enabled(False)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morig_amd import models, synth
nb = 8; dev = torch.device("cuda")
d = synth.make_batch_device(list(range(3000, 3000 + nb)), dev, n_side=64, n_pts=8192, with_skin=False)
d.num_graphs = nb
m = models.corrnet(input_feature=3, output_feature=64, temprature=0.07).train()
synth.load_recipe(m, 0, mild=True).to(dev)
def step():
    for p in m.parameters(): p.grad = None
    ov, op, vis, _ = m(d, True)
    ((ov[:, ::7] ** 2).mean() + (op[:, ::5] ** 2).mean() + (vis ** 2).mean()).backward()
with torch.enable_grad():
    step(); step(); torch.cuda.synchronize()
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"host enqueue {1e3*(t1-t0):.1f} ms; GPU drained {1e3*(t2-t0):.1f} ms")
    pr = cProfile.Profile(); pr.enable(); step(); pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:5000])
