"""Per-launch time distribution of the wide EdgeConv kernels (looks for rare slow launches). Usage: python tools/edge_jitter.py [reps]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from morig_amd import native, packing, synth
from morig_amd.native import Mat

def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    ops = native.get_ops()
    batch = synth.make_batch(range(16), n_side=64, with_skin=False).to("cuda")
    n = batch.pos.shape[0]
    csr = ops.csr_build(batch.geo_edge_index, n, pad4=True)
    torch.cuda.synchronize()
    csr.edge_count = int(csr.rowptr[-1].item())
    for H in (256, 128):
        g = torch.Generator().manual_seed(H)
        W = torch.randn(H, H, generator=g) / H ** 0.5
        ec = packing.to_device(packing.PackedEdge(H, None, None, W.contiguous(), torch.zeros(H), torch.ones(H), torch.zeros(H),
                                                  packing.split_f16(W.contiguous())), "cuda")
        ab = torch.randn(n * 5, 2 * H, device="cuda")
        out = torch.zeros(n * 5, H, device="cuda")
        fn = lambda: ops.edgeconv(Mat.of(ab, 0, H), Mat.of(ab, H, H), csr, ec, Mat.of(out), replicas=5, in_rep_stride=n, out_rep_stride=n)
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1))
        s = sorted(ts)
        med = s[len(s) // 2]
        slow = [(i, round(t, 2)) for i, t in enumerate(ts) if t > 1.5 * med]
        print(f"H={H}: min {s[0]:.3f} median {med:.3f} p95 {s[int(.95 * len(s))]:.3f} max {s[-1]:.3f} ms; launches > 1.5x median: {slow[:12]}")

main()
