"""Time of one JointNetMotion training step (forward / backward), default fp32-MFMA train forward vs MORIG_TRAIN_PRECISION=f16x3,
plus the per-op totals of the native ops (synchronised). usage: python tools/train_step_time.py [meshes]  (through gpurun)"""
import os, sys, time
from collections import defaultdict
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from morig_amd import models, native, synth  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = "cuda"
d = synth.make_batch(range(nb), n_side=64, with_skin=False).to(dev)
m = models.jointnet_motion(num_keyframes=5, chn_output=3, aggr_method="attn").train()
synth.load_recipe(m, 0, mild=True).to(dev)


def step(sync_split=False):
    for p in m.parameters():
        p.grad = None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    o = m(d, d.pred_flow)
    loss = (o[2] ** 2).mean() + (o[1] ** 2).mean()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    loss.backward()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) * 1e3, (t2 - t1) * 1e3


for prec in ("f32", "f16x3"):
    os.environ["MORIG_TRAIN_PRECISION"] = prec
    step(); step()
    f, b = zip(*[step() for _ in range(3)])
    print(f"train forward {prec}: {min(f):8.1f} ms   backward: {min(b):8.1f} ms   ({nb} meshes)")
os.environ["MORIG_TRAIN_PRECISION"] = "f32"
ops = native.get_ops()
agg, cnt = defaultdict(float), defaultdict(int)
for name in ("gemm", "gemm_tn", "edge_hidden", "edge_gather_relu", "col_stats", "col_affine", "segmax_affine_arg", "bn_backward_stats",
             "bn_relu_backward", "segmax_bn_backward_stats", "segmax_bn_relu_backward", "edge_scatter_backward", "csr_build"):
    fn = getattr(ops, name)

    def wrap(fn=fn, name=name):
        def inner(*a, **k):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r = fn(*a, **k)
            torch.cuda.synchronize(); agg[name] += (time.perf_counter() - t0) * 1e3; cnt[name] += 1
            return r
        return inner
    setattr(ops, name, wrap())
t0 = time.perf_counter(); step(); total = (time.perf_counter() - t0) * 1e3
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]):
    print(f"{k:28s} {v:9.1f} ms  {cnt[k]:5d} calls")
print(f"native ops {sum(agg.values()):.1f} ms of {total:.1f} ms (synchronised step)")
