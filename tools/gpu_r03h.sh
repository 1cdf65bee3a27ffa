#!/bin/bash
mkdir -p gpurun_out
TAG=${1:-r03h}
timeout 1500 python -m pytest tests/test_gpu_networks.py -q -m gpu -x --timeout=900 2>&1 | tail -5
bash tools/gpu_env_ab.sh pos_stream_$TAG MORIG_POS_STREAM 0 1 3
timeout 300 python tools/small_batch.py 1 8 graph 2>&1 | grep -v amdgpu | tee gpurun_out/small_batch_$TAG.txt
