#!/bin/bash
# boundary passes of the tpl / geo launch pairs in shared launches: tests, small-batch A/B by env (same box), headline A/B
cd /root/repo; mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_networks.py -m gpu -q -x > gpurun_out/r07i_tests.txt 2>&1; echo "tests: $?" > gpurun_out/r07i_rc.txt
: > gpurun_out/r07i_small.txt
for i in 1 2; do
timeout 600 python tools/fork_ab.py 1 2 4 8 2>&1 | tail -1 >> gpurun_out/r07i_small.txt
MORIG_EDGE_PAIR=0 timeout 600 python tools/fork_ab.py 1 2 4 8 2>&1 | tail -1 >> gpurun_out/r07i_small.txt
done
python - <<'PY'
import json
for l in open("gpurun_out/r07i_small.txt"):
    d=json.loads(l); print({k:(v["served_ms"], v["eager_ms"]) for k,v in d.items() if k.startswith("B")})
PY
for i in 1 2; do for e in 1 0; do
MORIG_EDGE_PAIR=$e python bench.py --steps 20 --warmup 5 --secondary 0 --cpu-seconds 0 2>/dev/null | tail -1 > gpurun_out/r07i_b.json
python - <<PY >> gpurun_out/r07i_rc.txt
import json; d=json.load(open("gpurun_out/r07i_b.json")); print("pair=$e", d["value"], d["ms_per_step"])
PY
done; done
cat gpurun_out/r07i_rc.txt; tail -3 gpurun_out/r07i_tests.txt; grep -n "FAILED\|Error" gpurun_out/r07i_tests.txt | head
