"""Which phase of the narrow (32-wide) EdgeConv tile costs what? Times morig_edgeconv_x3 on the headline geo / tpl graphs (5 replicas)
under MORIG_DEBUG_FLAGS (1 no epilogue, 2 no MFMA, 4 no gathers; read once per process: run one process per setting)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morig_amd import models, native, synth
from morig_amd.native import Mat

dev = torch.device("cuda:0")
nb = 16
d = synth.make_batch_device(range(nb), dev, n_side=64, with_skin=False)
m = synth.load_recipe(models.jointnet_motion(num_keyframes=5, chn_output=3, aggr_method="attn").eval(), 0, mild=True).to(dev)
ops = native.get_ops()
n = d.pos.shape[0]
pk = m.motionNet.gcu_1.packed(dev)
x = torch.zeros(5 * n, 4, device=dev); x[:, :3] = torch.randn(5 * n, 3, device=dev) * 0.05
for name, ei in (("tpl", d.tpl_edge_index), ("geo", d.geo_edge_index)):
    csr = ops.csr_build(ei, n)
    out = torch.zeros(5 * n, 96, device=dev)
    f = lambda: ops.edgeconv_x3(Mat.of(x), pk["x3g"], csr, pk["xg"], Mat.of(out, 0, 32, 0, n), replicas=5, in_rep_stride=n, out_rep_stride=n)
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize()
    rows = int(csr.rowptr[-1]) * 5
    ms = (time.perf_counter() - t0) / 20 * 1e3
    print(f"flags={os.environ.get('MORIG_DEBUG_FLAGS', '0')} {name}: {ms:.3f} ms for {rows} rows = {ms * 1e6 / (rows / 128):.0f} ns per 128-row tile / {256 * 4} concurrent", flush=True)
