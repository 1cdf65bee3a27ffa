import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from morig_amd import native, packing
from morig_amd.native import Mat
from emulate import EmuOps
import test_gpu_kernels as T
DEV = "cuda"
ops = native.get_ops()
for prec in ("f16x3", "f32"):
    ops.precision = prec
    for H in (64, 128, 256):
        n, e, hub, reps = 300, 2500, 5, 1
        g = torch.Generator().manual_seed(H + n)
        ei = T._rand_graph(n, e, 9, hub)
        ab = torch.randn(n, 2 * H + 4, generator=g)
        ec = T._edge_pack(H, 21)
        emu = EmuOps()
        cr = emu.csr_build(ei, n)
        ref = torch.zeros(n, H)
        emu.edgeconv(Mat.of(ab, 0, H), Mat.of(ab, H, H), cr, ec, Mat.of(ref))
        csr = ops.csr_build(ei.to(DEV), n)
        out = torch.zeros(n, H, device=DEV)
        abg = ab.to(DEV)
        ops.edgeconv(Mat.of(abg, 0, H), Mat.of(abg, H, H), csr, packing.to_device(ec, DEV), Mat.of(out))
        torch.cuda.synchronize()
        d = (out.cpu() - ref).abs()
        bad = (d > 1e-4).nonzero()
        rp = cr.rowptr.long()
        print(f"{prec} H={H}: bad elements {bad.shape[0]}; bad rows:", sorted(set(bad[:, 0].tolist()))[:20])
        for r in sorted(set(bad[:, 0].tolist()))[:6]:
            cols = bad[bad[:, 0] == r][:, 1].tolist()
            print(f"   node {r}: edges [{int(rp[r])},{int(rp[r+1])}) tile {int(rp[r])//128}..{(int(rp[r+1])-1)//128} deg {int(rp[r+1]-rp[r])} ncols {len(cols)} cols {cols[:8]}")
