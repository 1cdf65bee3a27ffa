"""rocprofv3 target: N eager forwards of jointnet_motion at B meshes (default 1) -- per-kernel durations of the small-batch operating point"""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from morig_amd import models, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda", 0)
d = bench.build_batch([1000 + i for i in range(B)], 64, dev=dev)
m = synth.load_recipe(models.jointnet_motion(num_keyframes=5, chn_output=3, aggr_method="attn").eval(), 0, mild=True).to(dev)
with torch.no_grad():
    for _ in range(N):
        m(d, d.pred_flow)
torch.cuda.synchronize()
