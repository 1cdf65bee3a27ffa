#!/bin/bash
# few-rows GEMM: kernel + network tests, small-batch timings, default bench line, per-launch time of the kernel
cd /root/repo; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_networks.py -m gpu -q -x > gpurun_out/r07a_tests.txt 2>&1; echo "tests: $?" > gpurun_out/r07a_rc.txt
timeout 600 python tools/fork_ab.py 1 2 8 2>&1 | tail -1 > gpurun_out/r07a_small.txt
MORIG_NO_FEW_ROWS=1 timeout 600 python tools/fork_ab.py 1 2 8 2>&1 | tail -1 >> gpurun_out/r07a_small.txt
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --secondary 0 --cpu-seconds 0 2>/dev/null | tail -1 > gpurun_out/r07a_b.json
python - <<PY >> gpurun_out/r07a_rc.txt
import json; d=json.load(open("gpurun_out/r07a_b.json")); print("new", d["value"], d["ms_per_step"])
PY
MORIG_NO_FEW_ROWS=1 python bench.py --steps 20 --warmup 5 --secondary 0 --cpu-seconds 0 2>/dev/null | tail -1 > gpurun_out/r07a_b0.json
python - <<PY >> gpurun_out/r07a_rc.txt
import json; d=json.load(open("gpurun_out/r07a_b0.json")); print("old", d["value"], d["ms_per_step"])
PY
done
bash tools/gpu_b1_timeline.sh r07a 1 > /dev/null 2>&1
grep -n "few_rows\|tile_kernel<128, 32, 0, 0, 1, false" gpurun_out/timeline_B1_r07a.txt | head
cat gpurun_out/r07a_rc.txt; cat gpurun_out/r07a_small.txt; tail -3 gpurun_out/r07a_tests.txt
