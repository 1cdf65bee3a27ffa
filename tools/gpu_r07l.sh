#!/bin/bash
# attention feed-forward pair through a reused hidden buffer of MORIG_FF_CHUNK rows (memory-side cache) vs one pass
cd /root/repo; mkdir -p gpurun_out; : > gpurun_out/r07l_rc.txt
for i in 1 2; do for c in 0 32768 65536 16384; do
MORIG_FF_CHUNK=$c python bench.py --steps 20 --warmup 5 --secondary 0 --cpu-seconds 0 2>/dev/null | tail -1 > gpurun_out/r07l_b.json
python - <<PY >> gpurun_out/r07l_rc.txt
import json; d=json.load(open("gpurun_out/r07l_b.json")); k=json.load(open("gpurun_out/bench_detail.json"))["kernels"]; print("chunk=$c", d["value"], d["ms_per_step"], {n: k[n]["ms_per_step"] for n in ("gemm_f16x3_bn128","gemm_f16x3_bn64")})
PY
done; done
cat gpurun_out/r07l_rc.txt
