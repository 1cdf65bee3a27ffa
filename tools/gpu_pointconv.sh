#!/bin/bash
# fused PointConv: kernel tests, corrnet / point-module network tests, per-op timeline and the corrnet bench with and without it
mkdir -p gpurun_out
TAG=${1:-a}
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "pointconv" --timeout=300 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_networks.py -q -m gpu -k "corrnet or point_modules or deformnet" --timeout=600 2>&1 | tail -8
timeout 300 python tools/op_timeline.py corrnet 32 2>&1 | grep -E "pointconv|edge_hidden|segmax|^sum|^fps|^gemm " 
for v in 1 0; do MORIG_POINTCONV_FUSED=$v timeout 600 python bench.py --workload corrnet --steps 30 --warmup 5 --cpu-seconds 0 --secondary 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fused=$v', d['value'], d['ms_per_step'], {k: v['ms_per_step'] for k, v in list(d['kernels'].items())[:6]})"; done
