#!/bin/bash
# H = 32 EdgeConv launches on the persistent kernel (gathered [A | B] rows): tests, per-launch times, bench A/B by env
cd /root/repo; mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_networks.py -m gpu -q -x > gpurun_out/r07k_tests.txt 2>&1; echo "tests: $?" > gpurun_out/r07k_rc.txt
bash tools/gpu_timeline.sh r07k jointnet cls_attention > /dev/null 2>&1
grep -n "edge_x3_kernel\|tile_kernel<32, 32, 1, 2, 1\|step span" gpurun_out/timeline_jointnet_r07k.txt | head -14
for w in jointnet corrnet; do for i in 1 2; do for e in 0 1; do
MORIG_X3_TILE=$e python bench.py --workload $w --steps 20 --warmup 5 --secondary 0 --cpu-seconds 0 2>/dev/null | tail -1 > gpurun_out/r07k_b.json
python - <<PY >> gpurun_out/r07k_rc.txt
import json; d=json.load(open("gpurun_out/r07k_b.json")); print("$w tile_engine=$e", d["value"], d["ms_per_step"])
PY
done; done; done
cat gpurun_out/r07k_rc.txt; tail -3 gpurun_out/r07k_tests.txt; grep -n "FAILED" gpurun_out/r07k_tests.txt | head
