"""Where does the transposed-accumulator GEMM put its results? X[m][k] = 128 m' + k (m' = m % 256), W = identity: Y[m][n] = 128 m' + n.
Prints the mismatching (m, n) and where their value came from."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morig_amd import native, packing
from morig_amd.native import Mat

DEV = "cuda"
o = native.get_ops()
o.precision = "f16x3"
M, N, K = int(sys.argv[1]) if len(sys.argv) > 1 else 300, 256, 256
x = torch.zeros(M, K)
for m in range(M):
    x[m] = 128.0 * (m % 256) + torch.arange(K).float() * 0.25
W = torch.eye(N, K)
lin = packing.pack_linear(W, torch.zeros(N))
xs = packing.split_f16(x).to(DEV)
for y_split in (False, True):
    if y_split:
        ys = torch.zeros(M, N, device=DEV)
        o.gemm(Mat.of(xs, 0, K), packing.to_device(lin, DEV), False, Y=Mat.of(ys), x_split=True, y_split=True)
        torch.cuda.synchronize()
        got = packing.unsplit_f16(ys.cpu(), N)
    else:
        y = torch.zeros(M, N, device=DEV)
        o.gemm(Mat.of(xs, 0, K), packing.to_device(lin, DEV), False, Y=Mat.of(y), x_split=True)
        torch.cuda.synchronize()
        got = y.cpu()
    want = x[:, :N]
    bad = (got - want).abs() > 1e-2
    print(f"y_split={y_split}: {int(bad.sum())} of {bad.numel()} wrong")
    idx = bad.nonzero()[:40]
    for m, n in idx.tolist():
        v = got[m, n].item()
        sm, sn = int(v // 128), (v % 128) / 0.25
        print(f"  Y[{m}][{n}] = {v:.2f}  (looks like X[{sm}][{sn:.0f}])")
    if bad.any():
        rows = bad.any(1).nonzero().flatten().tolist()
        cols = bad.any(0).nonzero().flatten().tolist()
        print("  bad rows:", rows[:64], "..." if len(rows) > 64 else "")
        print("  bad cols:", cols[:64], "..." if len(cols) > 64 else "")
