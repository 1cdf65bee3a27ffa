#!/bin/bash
mkdir -p gpurun_out
TAG=${1:-r03i}
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x --timeout=600 -k "csr or small_ops or vertex or rownorm or copy" 2>&1 | tail -4
timeout 1500 python -m pytest tests/test_gpu_networks.py -q -m gpu -x --timeout=900 2>&1 | tail -3
for rep in 1 2; do python bench.py --secondary 0 --cpu-seconds 0 --steps 30 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=r['kernels']
print(r['value'], r['ms_per_step_median'], ' '.join(f\"{n}={k[n]['ms_per_step']}\" for n in k if n in ('csr_build','copy','rownorm','cls_attention')), r['hbm_bound_kernels'])"; done | tee gpurun_out/csr_dual_$TAG.txt
