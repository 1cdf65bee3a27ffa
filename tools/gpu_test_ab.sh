#!/bin/bash
# parity first, then an environment A/B in ONE call: tools/gpu_test_ab.sh <tag> "<pytest -k expr for test_gpu_kernels>" <VAR> <A> <B> [reps]
mkdir -p gpurun_out
TAG=$1; KEXPR=$2; shift 2
timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x --timeout=900 -k "$KEXPR" 2>&1 | tail -6
timeout 1500 python -m pytest tests/test_gpu_networks.py -q -m gpu -x --timeout=900 2>&1 | tail -4
bash tools/gpu_env_ab.sh $TAG "$@"
