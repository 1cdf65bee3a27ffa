import os, sys, subprocess
code = '''
import torch, sys
from morig_amd import native
ops = native.get_ops()
g = torch.Generator().manual_seed(0)
for n in (64, 700, 2048, 5000, 8192):
    pos = torch.zeros(n, 4)
    pos[:, :3] = torch.rand(n, 3, generator=g)
    pos = pos.cuda()
    m = (n + 1) // 2
    ptr = torch.tensor([0, n], dtype=torch.int32).cuda(); optr = torch.tensor([0, m], dtype=torch.int32).cuda()
    idx = ops.fps(native.Mat.of(pos, 0, 3), ptr, optr, None, 1, n, m)
    torch.cuda.synchronize()
    print(n, idx[:10].tolist(), int(idx.long().sum()), len(set(idx.tolist())))
'''
for v in ("0", "1"):
    env = dict(os.environ, MORIG_FPS_BKT=v)
    print("BKT", v); sys.stdout.flush()
    subprocess.call([sys.executable, "-c", code], env=env)
