#!/bin/bash
# round 6: the exact path on bf16 x 6 (MORIG_SPLIT_BF16X6): whole GPU suite, then the training step A/B (MORIG_EXACT_ARITH=f32 | bf16x6)
mkdir -p gpurun_out
TAG=${1:-r06n}
timeout 1500 python -m pytest tests -q -m gpu -x --timeout=900 2>&1 | tail -40 > gpurun_out/pytest_gpu_$TAG.txt; tail -5 gpurun_out/pytest_gpu_$TAG.txt
for v in f32 bf16x6 f32 bf16x6; do
  MORIG_EXACT_ARITH=$v python tools/train_prof.py 8 6 2>&1 | tail -2 | sed "s/^/EXACT_ARITH=$v /" | tee -a gpurun_out/train_exact_ab_$TAG.txt
done
