#!/bin/bash
mkdir -p gpurun_out
TAG=${1:-r06u}
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x --timeout=600 -k "knn or point_kernels or fps" 2>&1 | tail -3 | tee gpurun_out/pytest_knn_$TAG.txt
timeout 900 python -m pytest tests/test_gpu_networks.py -q -m gpu -x --timeout=600 -k "corrnet or deformnet or point_modules" 2>&1 | tail -3 | tee -a gpurun_out/pytest_knn_$TAG.txt
bash tools/gpu_lib_sweep.sh ${TAG} 3 base_r06s
