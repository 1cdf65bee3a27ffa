#!/bin/bash
# register-resident CLS attention + row-major pack_tails: tests, per-launch times in the headline step's timeline, bench A/B by env where there is one
cd /root/repo; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_networks.py -m gpu -q -x > gpurun_out/r07b_tests.txt 2>&1; echo "tests: $?" > gpurun_out/r07b_rc.txt
bash tools/gpu_timeline.sh r07b jointnet cls_attention > /dev/null 2>&1
grep -n "cls_attention\|pack_tails\|few_rows\|step span\|idle" gpurun_out/timeline_jointnet_r07b.txt | head
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --secondary 0 --cpu-seconds 0 2>/dev/null | tail -1 > gpurun_out/r07b_b.json
python - <<PY >> gpurun_out/r07b_rc.txt
import json; d=json.load(open("gpurun_out/r07b_b.json")); print("new", d["value"], d["ms_per_step"])
PY
MORIG_NO_FEW_ROWS=1 MORIG_ATTN_LDS=1 python bench.py --steps 20 --warmup 5 --secondary 0 --cpu-seconds 0 2>/dev/null | tail -1 > gpurun_out/r07b_b0.json
python - <<PY >> gpurun_out/r07b_rc.txt
import json; d=json.load(open("gpurun_out/r07b_b0.json")); print("old", d["value"], d["ms_per_step"])
PY
done
cat gpurun_out/r07b_rc.txt; tail -3 gpurun_out/r07b_tests.txt
