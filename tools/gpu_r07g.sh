#!/bin/bash
# few-tile LDS-DMA launches on the 128 x 128 kernel: kernel + network tests, small-batch A/B by env (same box), headline sanity
cd /root/repo; mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_networks.py -m gpu -q -x > gpurun_out/r07g_tests.txt 2>&1; echo "tests: $?" > gpurun_out/r07g_rc.txt
: > gpurun_out/r07g_small.txt
for i in 1 2; do
timeout 600 python tools/fork_ab.py 1 2 4 8 2>&1 | tail -1 >> gpurun_out/r07g_small.txt
MORIG_DMA_SMALL_TILES=0 timeout 600 python tools/fork_ab.py 1 2 4 8 2>&1 | tail -1 >> gpurun_out/r07g_small.txt
done
MORIG_DMA_SMALL_TILES=64 timeout 600 python tools/fork_ab.py 1 2 4 8 2>&1 | tail -1 >> gpurun_out/r07g_small.txt
MORIG_DMA_SMALL_TILES=256 timeout 600 python tools/fork_ab.py 1 2 4 8 2>&1 | tail -1 >> gpurun_out/r07g_small.txt
python - <<'PY'
import json
for l in open("gpurun_out/r07g_small.txt"):
    d=json.loads(l); print({k:(v["served_ms"], v["eager_ms"]) for k,v in d.items() if k.startswith("B")})
PY
python bench.py --steps 20 --warmup 5 --secondary 0 --cpu-seconds 0 2>/dev/null | tail -1 > gpurun_out/r07g_b.json
python - <<PY >> gpurun_out/r07g_rc.txt
import json; d=json.load(open("gpurun_out/r07g_b.json")); print("bench", d["value"], d["ms_per_step"])
PY
cat gpurun_out/r07g_rc.txt; tail -3 gpurun_out/r07g_tests.txt
