"""Per-call timing of the native operators in ONE forward (every call synchronised: serial time, no stream overlap).
usage: python tools/op_timeline.py corrnet|deformnet|jointnet [pairs]   (run through gpurun)"""
import os
import sys
import time
from collections import defaultdict

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from morig_amd import models, native, synth          # noqa: E402

OPS = ["knn_search", "knn_apply", "gemm", "edgeconv", "edge_hidden", "segmax_gemm", "pointconv_fused", "fps", "ball_query", "csr_from_slots", "csr_build", "knn_interpolate",
       "gather_rows", "copy2d", "copy2d_pad", "cosine_nn", "rownorm"]


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "corrnet"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    dev = "cuda"
    pairs = wl in ("corrnet", "deformnet")
    data = synth.make_batch(range(n), n_side=64, with_skin=False, n_pts=8192 if pairs else 0).to(dev)
    if wl == "corrnet":
        model = models.corrnet(input_feature=3, output_feature=64, temprature=0.07).eval()
        run = lambda: model(data, True, False)
    elif wl == "deformnet":
        model = models.deformnet(tau_nce=0.07, num_interp=5).eval()
        run = lambda: model(data)
    else:
        model = models.jointnet_motion(num_keyframes=5, chn_output=3, aggr_method="attn").eval()
        run = lambda: model(data, data.pred_flow)
    synth.load_recipe(model, 0, mild=True).to(dev)
    os.environ["MORIG_SERIAL"] = "1"
    with torch.no_grad():
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        ops = native.get_ops()
        log = []
        for name in OPS:
            if not hasattr(ops, name):
                continue
            fn = getattr(ops, name)

            def wrap(fn=fn, name=name):
                def inner(*a, **k):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    r = fn(*a, **k)
                    torch.cuda.synchronize()
                    desc = []
                    for x in list(a) + list(k.values()):
                        if isinstance(x, native.Mat):
                            desc.append(f"{x.rows}x{x.cols}")
                        elif isinstance(x, torch.Tensor):
                            desc.append("t" + "x".join(str(d) for d in x.shape))
                        elif hasattr(x, "capacity"):
                            desc.append(f"E{x.capacity}")
                        elif hasattr(x, "N") and hasattr(x, "K"):
                            desc.append(f"K{x.K}N{x.N}")
                        elif hasattr(x, "H"):
                            desc.append(f"H{x.H}")
                    log.append((name, " ".join(desc), (time.perf_counter() - t0) * 1e3))
                    return r
                return inner
            setattr(ops, name, wrap())
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        total = (time.perf_counter() - t0) * 1e3
    agg = defaultdict(float)
    for name, desc, ms in log:
        print(f"{name:16s} {desc:48s} {ms:8.3f} ms")
        agg[name] += ms
    print("---- per op (serial, synchronised)")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1]):
        print(f"{k:16s} {v:8.3f} ms")
    print(f"sum {sum(agg.values()):.3f} ms, forward wall (with sync overhead) {total:.3f} ms")


if __name__ == "__main__":
    main()
