#!/bin/bash
# several whole-library variants (morig_amd/lib/variants/lib_<name>.so) against the regular build, interleaved: tools/gpu_lib_sweep.sh <tag> <reps> <name> ...
mkdir -p gpurun_out
TAG=$1; REPS=$2; shift 2
OUT=gpurun_out/lib_sweep_$TAG.txt; : > $OUT
for rep in $(seq $REPS); do for v in base "$@"; do
  if [ $v = base ]; then unset MORIG_HIP_LIB; else export MORIG_HIP_LIB=$PWD/morig_amd/lib/variants/lib_$v.so; fi
  python bench.py ${BENCH_ARGS:-} --secondary 0 --cpu-seconds 0 --steps 20 --warmup 5 --power-probe-seconds 0 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=json.load(open('gpurun_out/bench_detail.json')).get('kernels', {})
print('$v', r['value'], r['ms_per_step_median'], (r.get('roofline') or {}).get('sclk_under_load_mhz'), ' '.join(f\"{n}={k[n]['ms_per_step']}\" for n in list(k)[:5]))" | tee -a $OUT
done; done
unset MORIG_HIP_LIB
