"""Whole-network gradient conditioning: torch float32 vs float64 autograd on the oracle next to the HIP backward (the yardstick of
tests/test_gpu_backward.py::_grad_report). usage: python tools/debug_bw.py (through gpurun)"""
import copy, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_backward as T
from morig_amd import models, native, synth, train_backward as TB
from oracle import nets
DEV = "cuda"
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
n_side = int(sys.argv[2]) if len(sys.argv) > 2 else 9
native.get_ops().precision = prec
kw = dict(num_keyframes=5, chn_output=3, aggr_method="attn")
ref64 = T._randomise(nets.jointnet_motion(**kw), 7).train().double()
ref32 = copy.deepcopy(ref64).float()
mine = models.jointnet_motion(**kw).train(); mine.load_state_dict(copy.deepcopy(ref32.state_dict())); mine.to(DEV)
b = synth.make_batch(range(11, 13), n_side=n_side, with_skin=False)
g = torch.Generator().manual_seed(2); n = b.pos.shape[0]
w = [torch.randn(n, 5, 32, generator=g), torch.randn(n, 64, generator=g), torch.randn(n, 3, generator=g)]
def loss(o, dt, dev="cpu"): return sum((o[i] * w[i].to(dev, dt)).sum() for i in range(3))
for net, dt in ((ref64, torch.float64), (ref32, torch.float32)):
    bb = copy.copy(b); bb.pos = b.pos.to(dt)
    loss(net(bb, b.pred_flow.to(dt)), dt).backward()
bd = b.to(DEV)
loss(TB.motion_head_step(mine, bd, bd.pred_flow), torch.float32, DEV).backward()
rows = []
for (k, p), (_, q64), (_, q32) in zip(mine.named_parameters(), ref64.named_parameters(), ref32.named_parameters()):
    a, r, r32 = p.grad.detach().cpu().double().flatten(), q64.grad.flatten(), q32.grad.double().flatten()
    sc = max(float(r.abs().max()), 1e-12)
    rows.append((float((a - r).abs().max()) / sc, float((r32 - r).abs().max()) / sc, float(torch.dot(a, r) / (a.norm() * r.norm() + 1e-300)),
                 float(torch.dot(r32, r) / (r32.norm() * r.norm() + 1e-300)), k))
rows.sort(reverse=True)
print(prec, "n_side", n_side, "params", len(rows))
for e, e32, c, c32, k in rows[:8]: print(f"{e:9.2e} (fp32 ref {e32:9.2e})  cos {c:.6f} (fp32 ref {c32:.6f})  {k}")
print("worst cos mine", min(r[2] for r in rows), "worst cos fp32 ref", min(r[3] for r in rows))
print("median err mine", sorted(r[0] for r in rows)[len(rows)//2], "median err fp32 ref", sorted(r[1] for r in rows)[len(rows)//2])
