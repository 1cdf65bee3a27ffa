"""edge_ws<256> on the 4-aligned CSR vs the mixed-quad form on the MORIG_CSR_MIN4 CSR: one launch kind back to back for `seconds`
(sustained clocks), tpl and geo graphs of 16 meshes x 5 replicas, split rows out. usage (gpurun): python tools/mix_micro.py [seconds]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from morig_amd import native, packing, synth
from morig_amd.native import Mat
DEV = "cuda"
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5
ops = native.get_ops(); ops.precision = "f16x3"
batch = bench.build_batch(list(range(16)), 64, dev=torch.device("cuda", 0))
n, R, H = batch.pos.shape[0], 5, 256
g = torch.Generator().manual_seed(1)
W = (torch.randn(H, H, generator=g) / H ** 0.5).contiguous()
ec = packing.to_device(packing.PackedEdge(H, None, None, W, torch.zeros(H), torch.ones(H), torch.zeros(H), packing.split_f16(W)), DEV)
ab = torch.randn(R * n, 2 * H, device=DEV)
o = torch.empty(R * n, 2 * H + 32, device=DEV)
for gname in ("tpl", "geo"):
    ei = getattr(batch, gname + "_edge_index")
    for kind in ("pad4", "min4"):
        csr = ops.csr_build(ei, n, pad4=True) if kind == "pad4" else ops.csr_build(ei, n, min4=True)
        torch.cuda.synchronize()
        rows = int(csr.rowptr[-1].item())
        fn = lambda: ops.edgeconv(Mat.of(ab, 0, H), Mat.of(ab, H, H), csr, ec, Mat.of(o, 0, H), replicas=R, in_rep_stride=n, out_rep_stride=n, out_split=True)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        cs = bench.ClockSampler(period=0.02); cs.start()
        t0 = time.perf_counter(); k = 0
        while time.perf_counter() - t0 < secs:
            for _ in range(10):
                fn()
            torch.cuda.synchronize(); k += 10
        dt = time.perf_counter() - t0
        cs.stop(); c = cs.summary(t0, t0 + dt)
        print(f"{gname} {kind} dbg={os.environ.get('MORIG_DEBUG_FLAGS','0')} rows/rep {rows} ms/launch {dt/k*1e3:.4f} ns/row {dt/k/(rows*R)*1e9:.4f} clk {c.get('sclk_under_load_mhz')} W {c.get('socket_power_w')}", flush=True)
