import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from morig_amd import models, synth
DEV = "cuda"
m = models.corrnet(input_feature=3, output_feature=64, temprature=0.07).eval()
synth.load_recipe(m, 414, mild=False).to(DEV)
n_side, npts = int(os.environ.get("NS", "64")), int(os.environ.get("NP", "8192"))
n = n_side * n_side
def run(seeds):
    d = synth.make_batch(seeds, n_side=n_side, n_pts=npts).to(DEV)
    d.num_graphs = len(seeds)
    with torch.no_grad():
        ov, op, vis, _ = m(d, True, False)
    return ov, op, vis
ref = {s: run([s]) for s in (32, 3001)}
for B in (2, 3, 4, 8, 16, 32):
    for slot in sorted({0, 1, B // 2, B - 1}):
        seeds = [3000 + i for i in range(B)]
        seeds[slot] = 32
        ov, op, vis = run(seeds)
        dv = float((ov[slot * n:(slot + 1) * n] - ref[32][0]).abs().max())
        dp = float((op[slot * npts:(slot + 1) * npts] - ref[32][1]).abs().max())
        other = 1 if slot != 1 else 0
        do = float((op[other * npts:(other + 1) * npts] - ref[3001][1]).abs().max()) if seeds[other] == 3001 else -1
        print(f"B={B} slot={slot}: out_vtx diff {dv:.2e}  out_pts diff {dp:.2e}  (pair seed 3001 at {other}: {do:.2e})", flush=True)
