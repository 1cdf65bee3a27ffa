"""Which torch (non-native) operators a JointNetMotion training step issues, by operator and input shape, with their device time
(torch.profiler; the native operators appear as hip kernels without an aten parent and are left out). usage: through gpurun"""
import os, sys
import torch
from torch.profiler import profile, ProfilerActivity
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from morig_amd import models, synth  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
d = synth.make_batch(range(nb), n_side=64, with_skin=False).to("cuda")
m = models.jointnet_motion(num_keyframes=5, chn_output=3, aggr_method="attn").train()
synth.load_recipe(m, 0, mild=True).to("cuda")


def step():
    for p in m.parameters():
        p.grad = None
    o = m(d, d.pred_flow)
    ((o[2] ** 2).mean() + (o[1] ** 2).mean()).backward()


step(); step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=40, max_shapes_column_width=70))
