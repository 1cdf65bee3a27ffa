"""Debug aid: where does the DeformNet product path leave the oracle (features, mask, neighbour lists, flow_init)?"""
import sys
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from morig_amd import models, synth
from oracle import nets, pyg_primitives as P

DEV = torch.device("cuda:0")
kw = dict(tau_nce=0.07, num_interp=5)
ours = synth.load_recipe(models.deformnet(**kw).eval(), 61, mild=True)
ref = synth.load_recipe(nets.deformnet(**kw).eval(), 61, mild=True)
batch = synth.make_batch([91, 92], n_side=24, n_pts=2048)
torch.manual_seed(5)
with torch.no_grad():
    want = ref(batch)
torch.manual_seed(5)
got = [g.cpu() for g in ours.to(DEV)(batch.to(DEV))[:4]]
for name, g, w in zip(("pred_flow", "vtx_f", "pts_f", "vis"), got, want):
    d = (g - w).abs()
    print(f"{name:10s} max|d| {d.max().item():.3e} rows>1e-4: {(d.max(1)[0] > 1e-4).sum().item()} / {g.shape[0]}  scale {w.abs().max().item():.3f}")
# neighbour gap analysis on the oracle side
vf, pf, vis = want[1], want[2], want[3]
k = 5
yi, xi = P.knn(pf, vf, k + 1, batch.pts_batch, batch.vtx_batch, cosine=True)
sim = (pf[xi] * vf[yi]).sum(-1).view(-1, k + 1)
gap = (sim[:, k - 1] - sim[:, k])
print("min gap between 5th and 6th point similarity:", gap.min().item(), "rows with gap < 1e-6:", (gap < 1e-6).sum().item())
seen = (vis >= 0.5).squeeze(1); hid = ~seen
yi2, xi2 = P.knn(vf[seen], vf[hid], k + 1, batch.vtx_batch[seen], batch.vtx_batch[hid], cosine=True)
sim2 = (vf[seen][xi2] * vf[hid][yi2]).sum(-1).view(-1, k + 1)
gap2 = sim2[:, k - 1] - sim2[:, k]
print("min gap (visible neighbours of hidden vertices):", gap2.min().item(), "rows < 1e-6:", (gap2 < 1e-6).sum().item())
