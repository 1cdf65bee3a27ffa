#!/bin/bash
# FPS: threads per cloud 1024 / 512 / 256 -- bit-exact tests, then the corrnet workload under each
mkdir -p gpurun_out
TAG=${1:-r06j}
OUT=gpurun_out/fps_threads_$TAG.txt; : > $OUT
for T in 1024 512 256; do
  export MORIG_FPS_T=$T
  timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x --timeout=600 -k "fps or point_kernels" 2>&1 | tail -2 | tee -a $OUT
  for rep in 1 2; do
  python bench.py --workload corrnet --steps 20 --warmup 3 --cpu-seconds 0 --secondary 0 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=json.load(open('gpurun_out/bench_detail.json')).get('kernels', {})
print('FPS_T=$T', r['value'], r['ms_per_step'], ' '.join(f\"{n}={k[n]['ms_per_step']}\" for n in list(k)[:6]))" | tee -a $OUT
  done
done
