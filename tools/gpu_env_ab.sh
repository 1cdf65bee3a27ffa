#!/bin/bash
# A/B of one environment switch in ONE call: tools/gpu_env_ab.sh <tag> <VAR> <value A> <value B> [reps]   ("-" = unset)
mkdir -p gpurun_out
TAG=$1; VAR=$2; A=$3; B=$4; REPS=${5:-3}
OUT=gpurun_out/env_ab_$TAG.txt; : > $OUT
for rep in $(seq $REPS); do for v in "$A" "$B"; do
  if [ "$v" = "-" ]; then unset $VAR; else export $VAR="$v"; fi
  python bench.py --secondary 0 --cpu-seconds 0 --steps 30 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=json.load(open('gpurun_out/bench_detail.json')).get('kernels', {})
print('$VAR=$v', r['value'], r.get('ms_per_step_median', r.get('ms_per_step')), (r.get('roofline') or {}).get('sclk_under_load_mhz'), ' '.join(f\"{n}={k[n]['ms_per_step']}\" for n in list(k)[:8]))" | tee -a $OUT
done; done
