"""What clock and socket power does the chip hold while ONE operator runs back to back? (power-capped MI355X: a kernel's rate is
clock x work per cycle, and the clock follows the power the kernel draws.) usage (gpurun): python tools/op_clock.py [seconds]
Operators: edge H = 256 / 128 on the geo graph of 16 meshes x 5 replicas, the K = 1862 -> N = 1024 and K = 256 -> N = 1024 GEMMs."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                          # noqa: E402  (ClockSampler)
from morig_amd import native, packing, synth          # noqa: E402
from morig_amd.native import Mat                      # noqa: E402

DEV = "cuda"


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
    ops = native.get_ops()
    ops.precision = "f16x3"
    batch = synth.make_batch(range(16), n_side=64, with_skin=False).to(DEV)
    n, R = batch.pos.shape[0], 5
    csr = ops.csr_build(batch.geo_edge_index, n, pad4=True)
    torch.cuda.synchronize()
    csr.edge_count = int(csr.rowptr[-1].item())
    cases = {}
    for H in (256, 128):
        g = torch.Generator().manual_seed(H)
        W = (torch.randn(H, H, generator=g) / H ** 0.5).contiguous()
        ec = packing.to_device(packing.PackedEdge(H, None, None, W, torch.zeros(H), torch.ones(H), torch.zeros(H), packing.split_f16(W)), DEV)
        ab = torch.randn(R * n, 4 * H, device=DEV)
        o = torch.empty(R * n, 2 * H + 32, device=DEV)
        osp = os.environ.get("OP_SPLIT") == "1"            # results in the split-fp16 layout (the form the networks' wide units use)
        cases[f"edge_geo_H{H}"] = (lambda ab=ab, o=o, ec=ec, H=H: ops.edgeconv(Mat.of(ab, 0, H), Mat.of(ab, H, H), csr, ec, Mat.of(o, 0, H),
                                                                                 replicas=R, in_rep_stride=n, out_rep_stride=n, out_split=osp),
                                   2.0 * csr.edge_count * R * H * H)
    M = R * n
    for (K, N) in ((1862, 1024), (256, 1024), (64, 512), (544, 512), (288, 256)):
        g = torch.Generator().manual_seed(K)
        lin = packing.to_device(packing.pack_linear(torch.randn(N, K, generator=g) / K ** 0.5, torch.zeros(N)), DEV)
        x = torch.randn(M, (K + 31) // 32 * 32, device=DEV).half().float()      # (any finite bit pattern: timing and power only)
        y = torch.empty(M, N, device=DEV)
        cases[f"gemm_K{K}_N{N}"] = (lambda x=x, lin=lin, y=y, K=K: ops.gemm(Mat.of(x, 0, K), lin, True, Y=Mat.of(y), x_split=True, y_split=True),
                                    2.0 * M * K * N)
        if K in (544, 288):       # the unit MLP behind the EdgeConvs: fp32 rows in (tile kernel, in-kernel split) vs split rows in (LDS-DMA kernel)
            xf = torch.randn(M, K, device=DEV)
            cases[f"gemm_f32x_K{K}_N{N}"] = (lambda xf=xf, lin=lin, y=y, K=K: ops.gemm(Mat.of(xf, 0, K), lin, True, Y=Mat.of(y), y_split=True),
                                             2.0 * M * K * N)
    only = os.environ.get('OP_ONLY')
    for name, (fn, fl) in cases.items():
        if only and only not in name:
            continue
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        cs = bench.ClockSampler(period=0.02)
        cs.start()
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < secs:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            reps += 20
        dt = time.perf_counter() - t0
        cs.stop()
        st = cs.summary() if hasattr(cs, "summary") else {}
        print(f"{name:22s} {dt / reps * 1e3:8.3f} ms  {fl * reps / dt / 1e12:7.1f} TFLOP/s  sclk {st.get('sclk_under_load_mhz')} MHz "
              f"(p10 {st.get('sclk_p10_mhz')}, p90 {st.get('sclk_p90_mhz')})  power {st.get('socket_power_w')} W  samples {st.get('samples')}", flush=True)


if __name__ == "__main__":
    main()
