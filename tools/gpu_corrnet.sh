#!/bin/bash
# CorrNet / DeformNet: network parity tests, branch timing, the secondary bench lines
mkdir -p gpurun_out
TAG=${1:-a}
timeout 900 python -m pytest tests/test_gpu_networks.py -q -m gpu -k "corrnet or point_modules or deformnet or gcu" --timeout=600 2>&1 | tail -8
timeout 300 python tools/corrnet_branches.py 32 2>&1 | grep -v amdgpu
for rep in 1 2; do for v in 1 0; do MORIG_GEO_STREAM=$v timeout 600 python bench.py --workload corrnet --steps 30 --warmup 5 --cpu-seconds 0 --secondary 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('geo_stream=$v', d['value'], d['ms_per_step'], {k: v['ms_per_step'] for k, v in list(d['kernels'].items())[:7]})"; done; done
timeout 600 python bench.py --workload deformnet --steps 20 --warmup 5 --cpu-seconds 0 --secondary 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('deformnet', d['value'], d['ms_per_step'])"
