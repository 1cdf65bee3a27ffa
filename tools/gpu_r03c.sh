#!/bin/bash
# r03c: fp32-X GEMM tile 128 x 256 vs 128 x 128 (A/B in one call), per-launch shapes of the step
mkdir -p gpurun_out
TAG=${1:-r03c}
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x --timeout=600 -k "gemm" 2>&1 | tail -4
timeout 600 python tools/gemm_shapes.py 64 > gpurun_out/gemm_shapes_$TAG.txt 2>&1; head -40 gpurun_out/gemm_shapes_$TAG.txt
for rep in 1 2 3; do for v in 128 256; do
  MORIG_X32_BN=$v python bench.py --secondary 0 --cpu-seconds 0 --steps 30 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=r['kernels']; print('bn$v', r['value'], r['ms_per_step_median'], {n: (k[n]['ms_per_step'], k[n]['tflops']) for n in k if 'bn128' in n or 'bn256' in n})"
done; done | tee gpurun_out/x32_bn_ab_$TAG.txt
