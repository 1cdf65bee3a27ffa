#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_networks.py -m gpu -q -x > gpurun_out/r07c_tests.txt 2>&1; echo "tests: $?" > gpurun_out/r07c_rc.txt
bash tools/gpu_timeline.sh r07c jointnet cls_attention > /dev/null 2>&1
grep -n "cls_attention\|pack_tails\|few_rows\|copy2d_pad\|step span\|idle\|tile_kernel<128, 32, 0, 0, 1, false" gpurun_out/timeline_jointnet_r07c.txt | head -20
for i in 1 2 3; do
python bench.py --steps 20 --warmup 5 --secondary 0 --cpu-seconds 0 2>/dev/null | tail -1 > gpurun_out/r07c_b.json
python - <<PY >> gpurun_out/r07c_rc.txt
import json; d=json.load(open("gpurun_out/r07c_b.json")); print("new", d["value"], d["ms_per_step"])
PY
done
cat gpurun_out/r07c_rc.txt; tail -3 gpurun_out/r07c_tests.txt
