"""morig_gemm_tn (weight-gradient contraction, fp32 MFMA) on the shapes of a JointNetMotion training step. usage: through gpurun"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from morig_amd import native  # noqa: E402
from morig_amd.native import Mat  # noqa: E402

ops = native.get_ops()
for rows, N, K in [(32768, 1024, 1800), (32768, 1024, 900), (32768, 256, 1024), (230000, 256, 256), (500000, 128, 128), (230000, 16, 16), (32768, 512, 3)]:
    A = torch.randn(rows, N, device="cuda"); B = torch.randn(rows, (K + 3) // 4 * 4, device="cuda")
    for _ in range(3):
        ops.gemm_tn(Mat.of(A, 0, N), Mat.of(B, 0, K))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        ops.gemm_tn(Mat.of(A, 0, N), Mat.of(B, 0, K))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"gemm_tn rows={rows:7d} N={N:5d} K={K:5d}: {dt * 1e3:7.3f} ms  {2.0 * rows * N * K / dt / 1e12:6.1f} TFLOP/s")
