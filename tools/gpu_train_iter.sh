#!/bin/bash
# one iteration on the training step: the backward / training test modules, then the rocprofv3 kernel view of the step
# (tools/gpu_train_prof.sh). usage: tools/gpu_train_iter.sh <tag> [pytest -k expression]
TAG=${1:-it}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_backward.py tests/test_train_no_ties.py -q -m gpu -x --timeout=1200 ${2:+-k "$2"} 2>&1 | tail -6 | tee gpurun_out/pytest_train_$TAG.txt
bash tools/gpu_train_prof.sh $TAG 2>&1 | grep "train step\|kernel time" | tee gpurun_out/train_iter_$TAG.txt
