"""Is there a WELL-CONDITIONED training recipe for JointNetMotion? torch.autograd on the CPU oracle in float32 against
float64 with mild positive BatchNorm gains (0.8 .. 1.2), a normalised L2 loss and larger batches: per parameter tensor the max-norm and
the L2 error of the float32 gradient. Runs on the CPU: python tools/grad_conditioning.py"""
import copy, sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nets
from morig_amd import synth
torch.manual_seed(1234)
kw = dict(num_keyframes=5, chn_output=3, aggr_method="attn")
def mild(mod, seed, lo=0.8, hi=1.2):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in mod.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.weight.copy_(torch.rand(m.num_features, generator=g) * (hi - lo) + lo)
                m.bias.copy_(torch.randn(m.num_features, generator=g) * 0.1)
    return mod
for n_side, n_mesh in ((12, 3), (16, 2)):
    ref64 = mild(nets.jointnet_motion(**kw), 7).train().double()
    ref32 = copy.deepcopy(ref64).float()
    b = synth.make_batch(range(11, 11 + n_mesh), n_side=n_side)
    g = torch.Generator().manual_seed(2)
    n = b.pos.shape[0]
    tgt = torch.randn(n, 3, generator=g) * 0.1
    res = {}
    for name, net, dt in (("r64", ref64, torch.float64), ("r32", ref32, torch.float32)):
        bb = copy.copy(b); bb.pos = b.pos.to(dt)
        o = net(bb, b.pred_flow.to(dt))
        loss = ((o[2] - tgt.to(dt)) ** 2).mean() + 0.1 * (o[1] ** 2).mean()
        loss.backward()
        res[name] = float(loss.detach())
    worst, rels, scales = 0, [], []
    for (k, p64), (_, p32) in zip(ref64.named_parameters(), ref32.named_parameters()):
        if p64.grad is None: continue
        s = float(p64.grad.abs().max())
        e = float((p32.grad.double() - p64.grad).abs().max()) / max(s, 1e-12)
        rels.append((e, k, s))
    rels.sort(reverse=True)
    print(n_side, n_mesh, "loss", res, "worst fp32-vs-fp64 per tensor:", rels[:4], "median", rels[len(rels)//2][0], "grad scale range", min(r[2] for r in rels), max(r[2] for r in rels))
    l2 = []
    for (k, p64), (_, p32) in zip(ref64.named_parameters(), ref32.named_parameters()):
        if p64.grad is None: continue
        l2.append((float((p32.grad.double() - p64.grad).norm() / (p64.grad.norm() + 1e-300)), k))
    l2.sort(reverse=True)
    print("   relative L2 per tensor: worst", l2[:3], "median", l2[len(l2)//2][0])
    # where does it come from: count arg-max flips? forward outputs
