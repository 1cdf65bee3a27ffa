#!/bin/bash
# A/B of two library builds in ONE call (box-to-box spread is ~3 %): kernel + network parity on the new build, then alternating
# bench runs. usage: tools/gpu_ab_lib.sh <tag> <baseline .so> [pytest -k expression]
mkdir -p gpurun_out
TAG=${1:-ab}; BASE=${2:-morig_amd/lib/variants/lib_base.so}; KEXPR=${3:-gemm}
[ -n "$SKIP_TESTS" ] || timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x --timeout=900 -k "$KEXPR" 2>&1 | tail -15 > gpurun_out/pytest_kernels_$TAG.txt; tail -6 gpurun_out/pytest_kernels_$TAG.txt
[ -n "$SKIP_TESTS" ] || timeout 1500 python -m pytest tests/test_gpu_networks.py -q -m gpu -x --timeout=900 2>&1 | tail -40 > gpurun_out/pytest_networks_$TAG.txt; tail -4 gpurun_out/pytest_networks_$TAG.txt
OUT=gpurun_out/lib_ab_$TAG.txt; : > $OUT
for rep in 1 2 3; do for v in base new; do
  if [ $v = base ]; then export MORIG_HIP_LIB=$PWD/$BASE; else unset MORIG_HIP_LIB; fi
  python bench.py --secondary 0 --cpu-seconds 0 --steps 30 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=json.load(open('gpurun_out/bench_detail.json')).get('kernels', {})
print('$v', r['value'], r['ms_per_step_median'], (r.get('roofline') or {}).get('sclk_under_load_mhz'), ' '.join(f\"{n}={k[n]['ms_per_step']}\" for n in list(k)[:9]))" | tee -a $OUT
done; done
unset MORIG_HIP_LIB
