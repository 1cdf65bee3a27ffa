#!/bin/bash
mkdir -p gpurun_out
TAG=${1:-r03k}
timeout 900 python -m pytest tests/test_joints_host.py -q -m gpu --timeout=600 2>&1 | tail -4
timeout 600 python tools/bench_joints_batched.py 64 2>&1 | tail -2 | tee gpurun_out/joints_batched_stages_$TAG.txt
python - <<'PY' 2>&1 | tail -3 | tee -a gpurun_out/joints_batched_stages_$TAG.txt
import time, numpy as np, torch, sys
sys.path.insert(0, ".")
from morig_amd import joints as J
rng = np.random.default_rng(3); dev = torch.device("cuda:0"); B = 64
H, A = [], []
for _ in range(B):
    c = rng.uniform(-0.4, 0.4, (20, 3)); c[:, 0] = -np.abs(c[:, 0])
    H.append(c[rng.integers(0, 20, 4096)] + rng.normal(0, 0.03, (4096, 3))); A.append((rng.random((4096, 1)) ** 2).astype(np.float32))
jp = torch.from_numpy(np.concatenate(H)).to(dev); ja = torch.from_numpy(np.concatenate(A)).to(dev); jb = torch.arange(B, device=dev).repeat_interleave(4096)
for _ in range(2): J.extract_joints_batched(jp, ja, jb, None, 0.04, -1.0, 0.02, 30, num_graphs=B)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): J.extract_joints_batched(jp, ja, jb, None, 0.04, -1.0, 0.02, 30, num_graphs=B)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
print(f"extract_joints_batched end to end: {dt * 1e3:.1f} ms per 64 meshes = {B / dt:.0f} meshes/s")
PY
