"""Stage timings of the joint extraction on one MI355X: python tools/bench_joints.py [n_half]"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from morig_amd import joints as J

def main():
    n_half = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    rng = np.random.default_rng(3)
    centres = rng.uniform(-0.4, 0.4, (20, 3)); centres[:, 0] = -np.abs(centres[:, 0])
    half = centres[rng.integers(0, 20, n_half)] + rng.normal(0, 0.03, (n_half, 3))
    pts = np.concatenate([half, half * np.array([[-1, 1, 1]])])
    attn = np.tile((rng.random((n_half, 1)) ** 2).astype(np.float32), (2, 1))
    dev = torch.device("cuda:0")
    p = torch.from_numpy(pts).to(dev); a = torch.from_numpy(attn).to(dev)
    def timed(fn, reps=3):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3, r
    t_bw, bw = timed(lambda: J.estimate_bandwidth(p, 0.04))
    t_ms, modes = timed(lambda: J.meanshift_cluster(p, bw, a, 30))
    t_nms, kept = timed(lambda: J.nms_meanshift(modes, a, bw, 0.02))
    t_all, out = timed(lambda: J.extract_joints(p, a, None, 0.04, -1.0, 0.02, 30))
    print(f"n = {2 * n_half}: bandwidth {t_bw:.2f} ms, mean-shift (29 steps) {t_ms:.2f} ms, nms {t_nms:.2f} ms ({kept.shape[0]} kept), "
          f"extract_joints end to end {t_all:.2f} ms ({len(out['joints'])} joints)")
    # numpy reference timing of the same stages on the host (the oracle), small n only
    if n_half <= 2048:
        from oracle import joints as O
        t0 = time.perf_counter(); O.extract_joints(pts, attn, None, 0.04, -1.0, 0.02, 30); print(f"numpy oracle end to end: {(time.perf_counter() - t0) * 1e3:.0f} ms")

if __name__ == "__main__":
    main()
