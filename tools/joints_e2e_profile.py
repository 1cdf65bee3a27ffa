"""Where the end-to-end time of joints.extract_joints_batched goes (bench.py's secondary.joint_extraction workload): the call under
cProfile with HIP_LAUNCH_BLOCKING=1 (every launch synchronous, so device time lands on the host call that issued it), and the
unblocked wall time next to it.   usage: python tools/joints_e2e_profile.py [B]   (through gpurun)"""
import os, sys, time, cProfile, pstats, io
if "--unblocked" not in sys.argv:
    os.environ["HIP_LAUNCH_BLOCKING"] = "1"
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morig_amd import joints as J

args = [a for a in sys.argv[1:] if not a.startswith("--")]
B = int(args[0]) if args else 64
dev = torch.device("cuda:0")
rng = np.random.default_rng(3)
halves, attns = [], []
for _ in range(B):
    centres = rng.uniform(-0.4, 0.4, (20, 3)); centres[:, 0] = -np.abs(centres[:, 0])
    halves.append(centres[rng.integers(0, 20, 4096)] + rng.normal(0, 0.03, (4096, 3)))
    attns.append((rng.random((4096, 1)) ** 2).astype(np.float32))
jp = torch.from_numpy(np.concatenate(halves)).to(dev)
ja = torch.from_numpy(np.concatenate(attns)).to(dev)
jb = torch.arange(B, device=dev).repeat_interleave(4096)


def step():
    return J.extract_joints_batched(jp, ja, jb, None, 0.04, -1.0, 0.02, 30, num_graphs=B)


step(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 5 * 1e3
mode = "unblocked" if "--unblocked" in sys.argv else "HIP_LAUNCH_BLOCKING=1"
print(f"{mode}: {ms:.2f} ms per batch of {B} = {B / ms * 1e3:.0f} meshes/s")
if "--unblocked" not in sys.argv:
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        step()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
    print("(5 calls)")
    print("\n".join(l for l in s.getvalue().splitlines() if l.strip())[:6000])
