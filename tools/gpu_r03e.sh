#!/bin/bash
# r03e: batched joint extraction tests + nontemporal-store A/B + bench line
mkdir -p gpurun_out
TAG=${1:-r03e}
timeout 900 python -m pytest tests/test_joints_host.py -q -m gpu --timeout=600 2>&1 | tail -8
OUT=gpurun_out/epi_nt_ab_$TAG.txt; : > $OUT
for rep in 1 2 3; do for v in plain nt; do
  if [ $v = nt ]; then export MORIG_HIP_LIB=$PWD/morig_amd/lib/variants/lib_epi_nt.so; else unset MORIG_HIP_LIB; fi
  python bench.py --secondary 0 --cpu-seconds 0 --steps 30 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=r['kernels']
print('$v', r['value'], r['ms_per_step_median'], ' '.join(f\"{n}={k[n]['ms_per_step']}\" for n in list(k)[:6]))" | tee -a $OUT
done; done
unset MORIG_HIP_LIB
python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
python -c "
import json; r=json.loads(open('gpurun_out/bench_$TAG.json').read().strip().splitlines()[-1])
print(r['value'], r['ms_per_step']); print(json.dumps(r['secondary']['joint_extraction'], indent=0)); print(json.dumps(r['roofline'], indent=0)[:1500])"; tail -3 gpurun_out/bench_$TAG.err
