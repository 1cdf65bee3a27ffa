#!/bin/bash
mkdir -p gpurun_out
TAG=${1:-r06e}
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x --timeout=600 -k "csr_build or edgeconv_split" 2>&1 | tail -15 > gpurun_out/pytest_kernels_$TAG.txt; tail -6 gpurun_out/pytest_kernels_$TAG.txt
: > gpurun_out/mix_micro_$TAG.txt
for d in ${DBGS:-0 1}; do MORIG_DEBUG_FLAGS=$d python tools/mix_micro.py 1.5 2>/dev/null | tee -a gpurun_out/mix_micro_$TAG.txt; done
