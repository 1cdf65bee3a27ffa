"""N JointNetMotion training steps (forward + backward), for `rocprofv3 --kernel-trace --stats` (tools/gpu_train_prof.sh)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from morig_amd import models, synth  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
d = synth.make_batch(range(nb), n_side=64, with_skin=False).to("cuda")
m = models.jointnet_motion(num_keyframes=5, chn_output=3, aggr_method="attn").train()
synth.load_recipe(m, 0, mild=True).to("cuda")


def step():
    for p in m.parameters():
        p.grad = None
    o = m(d, d.pred_flow)
    ((o[2] ** 2).mean() + (o[1] ** 2).mean()).backward()


step(); step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
print(f"train step: {(time.perf_counter() - t0) / steps * 1e3:.1f} ms ({nb} meshes, {steps} steps)")
