#!/bin/bash
# what the re-run behind the range guard costs: the headline workload on the EXACT path (MORIG_PRECISION=f32), fp32 MFMA vs bf16 x 6
mkdir -p gpurun_out
TAG=${1:-r06p}
OUT=gpurun_out/exact_path_$TAG.txt; : > $OUT
for v in f32 bf16x6; do
  MORIG_PRECISION=f32 MORIG_EXACT_ARITH=$v python bench.py --steps 6 --warmup 2 --cpu-seconds 0 --secondary 0 --prof-steps 2 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=json.load(open('gpurun_out/bench_detail.json')).get('kernels', {})
print('PRECISION=f32 EXACT_ARITH=$v', r['value'], r['ms_per_step'], ' '.join(f\"{n}={k[n]['ms_per_step']}\" for n in list(k)[:6]))" | tee -a $OUT
done
python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --secondary 0 --prof-steps 0 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default (f16x3)', r['value'], r['ms_per_step'])" | tee -a $OUT
