"""Power and clock of the matrix pipe alone (tools/ubench/mfma_power.hip): random operands / zeros / + LDS fragment reads / + VALU.
usage (gpurun): python tools/mfma_power.py [seconds]. Prints the MFMA rate (dense f16 TFLOP/s), clock, socket power per mode."""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                          # noqa: E402

lib = ctypes.CDLL(os.path.join(ROOT, "tools", "ubench", "libmfma_power.so"))
lib.mfma_power_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
out = torch.zeros(16, device="cuda")
ITERS = 20000
names = {0: "random operands", 1: "zero operands", 2: "random + 12 ds_read_b128 / 24 MFMA", 3: "random + 12 v_fma / 24 MFMA"}
for mode in (0, 1, 2, 3, 0):
    lib.mfma_power_run(mode, ITERS, 1, out.data_ptr())
    cs = bench.ClockSampler(period=0.02).start()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < secs:
        lib.mfma_power_run(mode, ITERS, 4, out.data_ptr()); n += 4
    dt = time.perf_counter() - t0
    cs.stop(); st = cs.summary()
    flops = 256 * 8 * ITERS * 24 * 2.0 * 32 * 32 * 16 * n
    print(f"mode {mode} {names[mode]:38s} {flops / dt / 1e12:8.1f} TFLOP/s (f16 MFMA)  sclk {st['sclk_under_load_mhz']} MHz  power {st['socket_power_w']} W", flush=True)
