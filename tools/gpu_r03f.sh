#!/bin/bash
# r03f: grouped position branches A/B + network parity, batched joint-extraction stage timings
mkdir -p gpurun_out
TAG=${1:-r03f}
timeout 1500 python -m pytest tests/test_gpu_networks.py -q -m gpu -x --timeout=900 2>&1 | tail -8 > gpurun_out/pytest_networks_$TAG.txt; tail -4 gpurun_out/pytest_networks_$TAG.txt
OUT=gpurun_out/pos_groups_ab_$TAG.txt; : > $OUT
for rep in 1 2 3; do for v in 0 1; do
  MORIG_POS_GROUPS=$v python bench.py --secondary 0 --cpu-seconds 0 --steps 30 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=r['kernels']
print('groups=$v', r['value'], r['ms_per_step_median'], ' '.join(f\"{n}={k[n]['ms_per_step']}\" for n in k if n in ('edgeconv_h16','edgeconv_f16x3_h32','copy','gemm_f16x3_bn64','gemm_f16x3_bn32','gemm_f16x3_bn128')))" | tee -a $OUT
done; done
timeout 600 python tools/bench_joints_batched.py 64 2>&1 | tail -3 | tee gpurun_out/joints_batched_stages_$TAG.txt
