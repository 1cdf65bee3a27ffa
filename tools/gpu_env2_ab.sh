#!/bin/bash
# A/B of two environment settings in ONE call: tools/gpu_env2_ab.sh <tag> "<VAR=val ...>" "<VAR=val ...>" [reps]   ("-" = nothing set)
mkdir -p gpurun_out
TAG=$1; A=$2; B=$3; REPS=${4:-3}
OUT=gpurun_out/env_ab_$TAG.txt; : > $OUT
for rep in $(seq $REPS); do for v in "$A" "$B"; do
  if [ "$v" = "-" ]; then envs=""; else envs="$v"; fi
  env $envs python bench.py --secondary 0 --cpu-seconds 0 --steps 30 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=json.load(open('gpurun_out/bench_detail.json')).get('kernels', {})
print('[$v]', r['value'], r.get('ms_per_step_median', r.get('ms_per_step')), (r.get('roofline') or {}).get('sclk_under_load_mhz'), ' '.join(f\"{n}={k[n]['ms_per_step']}\" for n in list(k)[:8]))" | tee -a $OUT
done; done
