"""Stage timings of the BATCHED joint extraction (64 meshes x 4096 shifted points + mirror images): python tools/bench_joints_batched.py [B]"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from morig_amd import joints as J, native

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
rng = np.random.default_rng(3)
dev = torch.device("cuda:0")
P, A = [], []
for _ in range(B):
    centres = rng.uniform(-0.4, 0.4, (20, 3)); centres[:, 0] = -np.abs(centres[:, 0])
    h = centres[rng.integers(0, 20, 4096)] + rng.normal(0, 0.03, (4096, 3))
    a = (rng.random((4096, 1)) ** 2).astype(np.float32)
    P.append(np.concatenate([h, h * np.array([[-1, 1, 1]])])); A.append(np.tile(a, (2, 1)))
pts = torch.from_numpy(np.concatenate(P)).to(dev)
att = torch.from_numpy(np.concatenate(A)).to(dev).reshape(-1).contiguous()
ptr = torch.arange(B + 1, dtype=torch.int32, device=dev) * 8192
ops = native.get_ops()


def timed(fn, reps=3):
    r = fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, r


t_bw, bw = timed(lambda: ops.knn_bandwidth_batched(pts, ptr, 8192, 0.04))
t_ms0, modes0 = timed(lambda: ops.meanshift_batched(pts, att, ptr, 8192, bw, 30))
t_ms, modes = timed(lambda: ops.meanshift_batched_sorted(pts, att, ptr, 8192, bw, 30))
print(f"mean-shift plain {t_ms0:.2f} ms, Morton-sorted + culled {t_ms:.2f} ms, max |diff| {float((modes - modes0).abs().max()):.2e}")
t_cnt, counts = timed(lambda: ops.nms_counts_batched(modes, ptr, 8192, bw))
t_msc, (modes_c, counts_c) = timed(lambda: ops.meanshift_batched_sorted(pts, att, ptr, 8192, bw, 30, with_counts=True))
print(f"neighbour counts: plain kernel {t_cnt:.2f} ms; in sorted order with box culling +{t_msc - t_ms:.2f} ms on top of the mean-shift; "
      f"equal: {bool((counts_c == counts).all())}")
ch = counts.cpu().numpy().astype(np.int64)
t0 = time.perf_counter()
order = np.concatenate([np.argsort(ch[b * 8192:(b + 1) * 8192])[::-1] for b in range(B)]).astype(np.int32)
t_sort = (time.perf_counter() - t0) * 1e3
od = torch.from_numpy(order).to(dev)
t_gr, alive = timed(lambda: ops.nms_greedy_batched(modes, att, ptr, bw, od, 0.02, 0.7))
print(f"B = {B}: bandwidth {t_bw:.2f} ms | mean-shift 29 steps {t_ms:.2f} ms | nms counts {t_cnt:.2f} ms | host argsort {t_sort:.2f} ms | "
      f"greedy {t_gr:.2f} ms | survivors/mesh {float(alive.sum()) / B:.1f}")
