"""Is a workload launch-bound? Host time to ENQUEUE one forward (no synchronisation inside the loop) against the GPU-paced step time.
usage: python tools/host_enqueue_time.py [corrnet|deformnet|jointnet] (through gpurun)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "corrnet"
pairs = wl in ("corrnet", "deformnet")
B = 32 if pairs else 64
dev = torch.device("cuda", 0)
data = bench.build_batch([1000 + i for i in range(B)], 64, with_skin=False, n_pts=8192 if pairs else 0).to(dev)
step = bench.make_step(wl, data, dev, lambda t: t)
from morig_amd import native  # noqa: E402
ops = native.get_ops()
enq = []
_guarded = ops.guarded


def guarded(device, fn, rerun=True):                 # time the enqueue part of a forward: fn() without the flag read behind it
    def timed():
        a = time.perf_counter()
        try:
            return fn()
        finally:
            enq.append(time.perf_counter() - a)
    return _guarded(device, timed, rerun)


ops.guarded = guarded
with torch.no_grad():
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    N = 30
    host = []
    t0 = time.perf_counter()
    for _ in range(N):
        a = time.perf_counter(); step(); host.append(time.perf_counter() - a)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    # the same with a synchronisation after every forward: host enqueue time when the queue is empty
    solo = []
    for _ in range(10):
        torch.cuda.synchronize(); a = time.perf_counter(); step(); solo.append(time.perf_counter() - a)
    torch.cuda.synchronize()
host.sort(); solo.sort()
e = sorted(enq[-(N + 10):-10])
print(f"{wl}: pure host enqueue time of one forward (inside the guard, before the flag read): median {e[len(e) // 2] * 1e3:.2f} ms, min {e[0] * 1e3:.2f} ms")
print(f"{wl}: GPU-paced step {t_all / N * 1e3:.2f} ms; host enqueue loop {t_enq / N * 1e3:.2f} ms per forward "
      f"(median call {host[N // 2] * 1e3:.2f} ms); enqueue on an empty queue: median {solo[5] * 1e3:.2f} ms, min {solo[0] * 1e3:.2f} ms")
