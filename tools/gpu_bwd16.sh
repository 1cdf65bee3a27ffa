#!/bin/bash
# bf16 x 3 backward contractions: parity tests of the backward suite on both settings, the weight-gradient GEMM on the
# training step's shapes (MORIG_TRAIN_BWD=f32 vs default), the training step time. usage: tools/gpu_bwd16.sh <tag>
TAG=${1:-bwd16}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_backward.py tests/test_train_no_ties.py -q -m gpu --timeout=1200 2>&1 | tail -12 | tee gpurun_out/pytest_bwd_$TAG.txt
for v in f32 bf16x3; do echo "== MORIG_TRAIN_BWD=$v"; MORIG_TRAIN_BWD=$v python tools/gemm_tn_bench.py 2>&1 | grep gemm_tn; done | tee gpurun_out/gemm_tn_$TAG.txt
for v in f32 bf16x3; do echo "== MORIG_TRAIN_BWD=$v"; MORIG_TRAIN_BWD=$v python tools/train_step_time.py 8 2>&1 | grep "train forward\|gemm_tn \|gemm  \|native ops"; done | tee gpurun_out/train_step_$TAG.txt
