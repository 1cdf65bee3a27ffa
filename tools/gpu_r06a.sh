#!/bin/bash
# round 6, first call: the new ABI-3 library -- kernel tests of the K tail, the whole network suite, then the A/B of the K-tail plan
mkdir -p gpurun_out
TAG=${1:-r06a}
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x --timeout=600 -k "k_tail or pack_tails or gemm or copy2d or edgeconv_split" 2>&1 | tail -8 > gpurun_out/pytest_kernels_$TAG.txt; tail -4 gpurun_out/pytest_kernels_$TAG.txt
timeout 1500 python -m pytest tests/test_gpu_networks.py -q -m gpu -x --timeout=900 2>&1 | tail -60 > gpurun_out/pytest_networks_$TAG.txt; tail -4 gpurun_out/pytest_networks_$TAG.txt
bash tools/gpu_env_ab.sh ${TAG}_tail MORIG_GEMM_TAIL 0 1 3
