#!/bin/bash
mkdir -p gpurun_out
TAG=${1:-r03g}
timeout 900 python -m pytest tests/test_joints_host.py -q -m gpu --timeout=600 2>&1 | tail -8
timeout 600 python tools/bench_joints_batched.py 64 2>&1 | tail -3 | tee gpurun_out/joints_batched_stages_$TAG.txt
bash tools/gpu_env_ab.sh dma_persist_$TAG MORIG_DMA_PERSIST - 1 2
