#!/bin/bash
# ping-pong GEMM (gemm_pp.hip): its parity test alone under a short timeout (a barrier bug would hang), then the GEMM kernel tests and
# the network goldens with MORIG_GEMM_PP=1, then the A/B of the switch (alternating bench runs in ONE call). usage: tools/gpu_pp.sh <tag>
TAG=${1:-pp}
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x --timeout=240 -k "ping_pong" 2>&1 | tail -15 > gpurun_out/pytest_pp_$TAG.txt; tail -8 gpurun_out/pytest_pp_$TAG.txt
if ! grep -q " passed" gpurun_out/pytest_pp_$TAG.txt || grep -q "failed\|error" gpurun_out/pytest_pp_$TAG.txt; then echo "ping-pong parity test did not pass: stopping"; exit 1; fi
MORIG_GEMM_PP=1 timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x --timeout=600 -k "gemm" 2>&1 | tail -6
MORIG_GEMM_PP=1 timeout 1200 python -m pytest tests/test_gpu_networks.py -q -m gpu -x --timeout=900 2>&1 | tail -4
bash tools/gpu_env_ab.sh $TAG MORIG_GEMM_PP 0 1 3
