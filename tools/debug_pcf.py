import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from morig_amd import native, packing
from morig_amd.native import Mat
from emulate import EmuOps
import test_gpu_kernels as T
torch.set_printoptions(linewidth=200, precision=4, sci_mode=False)
ops = native.get_ops()
H, H3, cx = int(sys.argv[1]), int(sys.argv[2]), 0
mode = sys.argv[3] if len(sys.argv) > 3 else "rand"
nn_ = T._pointconv_module(cx, H, H3, 3)
parts = mode.split("+")
if mode != "rand":
    ref = T._pointconv_module(cx, H, H3, 3)
    with torch.no_grad():
        for i, l in enumerate(nn_):
            l[0].weight.zero_(); l[0].bias.zero_()
            n = min(l[0].weight.shape)
            l[0].weight[:n, :n] = torch.eye(n)
            if i == 2 and H3 > H:
                l[0].weight[H:, :H] = torch.eye(H)[: H3 - H]
            l[2].weight.fill_(1.0); l[2].bias.zero_(); l[2].running_mean.zero_(); l[2].running_var.fill_(1.0 - l[2].eps)
            if f"w{i + 1}" in parts: l[0].weight.copy_(ref[i][0].weight)
            if f"b{i + 1}" in parts: l[0].bias.copy_(ref[i][0].bias)
            if f"bn{i + 1}" in parts:
                l[2].weight.copy_(ref[i][2].weight); l[2].bias.copy_(ref[i][2].bias)
                l[2].running_mean.copy_(ref[i][2].running_mean); l[2].running_var.copy_(ref[i][2].running_var)
pk = packing.pack_pointconv(nn_, cx)
n_src, n_ctr = 200, 64
g = torch.Generator().manual_seed(0)
slots = torch.randint(0, n_src, (n_ctr, 64), generator=g)
if mode != "rand" and "slots" not in parts:
    slots[:] = torch.arange(n_ctr)[:, None]          # every edge = the self loop: out[c] = message(c, c)
coo = torch.stack([slots.reshape(-1), torch.arange(n_ctr).repeat_interleave(64)])
A, Bm = torch.randn(n_ctr, H, generator=g), torch.randn(n_src, H, generator=g)
if mode != "rand" and "A" not in parts:
    A.zero_()
if mode != "rand" and "B" not in parts:
    Bm = torch.arange(n_src)[:, None] * 1.0 + torch.arange(H)[None, :] * 0.01 + 1.0
emu = EmuOps(); emu.emulate_split = True
want = torch.zeros(n_ctr, H3)
emu.pointconv_fused(Mat.of(A), Mat.of(Bm), coo, 64, pk, Mat.of(want))
pkd = packing.to_device(pk, "cuda")
got = torch.zeros(n_ctr, H3, device="cuda")
ops.pointconv_fused(Mat.of(A.cuda()), Mat.of(Bm.cuda()), coo.cuda(), 64, pkd, Mat.of(got))
torch.cuda.synchronize()
err = (got.cpu() - want).abs()
print(mode, "max err", err.max().item())
if err.max() > 1e-3: print("err per column:", (err.max(0).values > 1e-3).int().tolist())
if err.max() > 1e-3: print("err per row   :", (err.max(1).values > 1e-3).int().tolist())
if False:
    print("want[3,:16]", want[3, :16]); print("got [3,:16]", got[3, :16].cpu())
    print("want[3,H:H+16]", want[3, H:H + 16]); print("got [3,H:H+16]", got[3, H:H + 16].cpu())
