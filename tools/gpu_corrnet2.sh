#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "knn or ball or fps or pointconv or cosine" --timeout=300 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_networks.py -q -m gpu -k "corrnet or point_modules or deformnet or gcu" --timeout=600 2>&1 | tail -3
timeout 300 python tools/op_timeline.py corrnet 32 2>&1 | grep -E "^knn|^ball|^sum|^cosine|^csr|^copy|^gemm  |^edgeconv  |^fps  "
for rep in 1 2 3; do timeout 600 python bench.py --workload corrnet --steps 30 --warmup 5 --cpu-seconds 0 --secondary 0 --prof-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('corrnet', d['value'], d['ms_per_step'], d['ms_per_step_median'])"; done
