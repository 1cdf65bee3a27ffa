#!/bin/bash
# round 6: the mixed-quad EdgeConv (MORIG_CSR_MIN4 + edge_ws<256, true, true>): kernel tests, network goldens, then the A/B against the 4-aligned CSRs
mkdir -p gpurun_out
TAG=${1:-r06d}
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x --timeout=600 -k "csr_build or edgeconv_split" 2>&1 | tail -15 > gpurun_out/pytest_kernels_$TAG.txt; tail -6 gpurun_out/pytest_kernels_$TAG.txt
timeout 1500 python -m pytest tests/test_gpu_networks.py -q -m gpu -x --timeout=900 2>&1 | tail -60 > gpurun_out/pytest_networks_$TAG.txt; tail -4 gpurun_out/pytest_networks_$TAG.txt
bash tools/gpu_env_ab.sh ${TAG}_mix MORIG_EDGE_MIX 0 1 3
