"""GPU micro-benchmark of the two hot operators at the headline shapes (run through gpurun).
MORIG_DEBUG_FLAGS ablates kernel phases: 1 no epilogue, 2 no MFMA, 4 no gather loads, 8 no W loads, 16 stage once."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from morig_amd import native, packing, synth          # noqa: E402
from morig_amd.native import Mat                      # noqa: E402

DEV = "cuda"


def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
    nmesh = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    ops = native.get_ops()
    ops.precision = prec
    batch = synth.make_batch(range(nmesh), n_side=64, with_skin=False).to(DEV)
    n = batch.pos.shape[0]
    R = 5
    out = {}
    for gname, ei in (() if os.environ.get("MB_NOEDGE") else (("tpl", batch.tpl_edge_index), ("geo", batch.geo_edge_index))):
        E = int(ops.csr_build(ei, n).rowptr[-1].item())          # algorithmic edge count (no padding)
        csr = ops.csr_build(ei, n, pad4=os.environ.get("MB_PAD4", "1") == "1")
        torch.cuda.synchronize()
        csr.edge_count = int(csr.rowptr[-1].item())
        for H in [int(h) for h in os.environ.get("MB_HS", "256,128").split(",")]:
            g = torch.Generator().manual_seed(H)
            Hp = max(H, 32)
            W = torch.zeros(Hp, Hp)
            W[:H, :H] = torch.randn(H, H, generator=g) / H ** 0.5
            ec = packing.PackedEdge(H, None, None, W.contiguous(), torch.zeros(Hp), torch.ones(Hp), torch.zeros(Hp),
                                    packing.split_f16(W.contiguous()) if H >= 32 else None)
            ec = packing.to_device(ec, DEV)
            ab = torch.randn(R * n, 4 * H, device=DEV)
            o = torch.empty(R * n, 2 * H + 32, device=DEV)
            ms = timed(lambda: ops.edgeconv(Mat.of(ab, 0, H), Mat.of(ab, H, H), csr, ec, Mat.of(o, 0, H), replicas=R,
                                            in_rep_stride=n, out_rep_stride=n))
            fl = 2.0 * E * R * H * H
            out[f"edge_{gname}_H{H}"] = (ms, fl / ms / 1e9)
    M = R * n
    for (K, N) in (() if os.environ.get("MB_NOGEMM") else ((1862, 1024), (544, 512), (256, 1024), (1024, 256), (832, 1024))):
        g = torch.Generator().manual_seed(K)
        lin = packing.to_device(packing.pack_linear(torch.randn(N, K, generator=g) / K ** 0.5, torch.zeros(N)), DEV)
        x = torch.randn(M, (K + 3) // 4 * 4, device=DEV)
        y = torch.empty(M, N, device=DEV)
        ms = timed(lambda: ops.gemm(Mat.of(x, 0, K), lin, True, Y=Mat.of(y)), reps=3)
        out[f"gemm_M{M}_K{K}_N{N}"] = (ms, 2.0 * M * K * N / ms / 1e9)
        if prec == "f16x3":
            xs = torch.randn(M, (K + 31) // 32 * 32, device=DEV)          # bit pattern irrelevant for timing? no: use a real split
            xs = packing.split_f16(xs.cpu()).to(DEV) if M * K < 2e8 else xs.half().float()
            ms = timed(lambda: ops.gemm(Mat.of(xs, 0, K), lin, True, Y=Mat.of(y), x_split=True, y_split=True), reps=3)
            out[f"gemm16_M{M}_K{K}_N{N}"] = (ms, 2.0 * M * K * N / ms / 1e9)
    flags = os.environ.get("MORIG_DEBUG_FLAGS", "0")
    for k, (ms, tf) in out.items():
        print(f"prec={prec} dbg={flags:>2} {k:28s} {ms:9.3f} ms  {tf:8.1f} TFLOP/s")


if __name__ == "__main__":
    main()
