#!/bin/bash
# joint-extraction parity tests, then the stage timings with one environment switch at two values in ONE call:
#   tools/gpu_joints_ab.sh <tag> <VAR> <A> <B>      ("-" = unset)
mkdir -p gpurun_out
TAG=$1; VAR=$2; A=$3; B=$4
timeout 900 python -m pytest tests/test_joints_host.py tests/test_formats.py -q -m gpu -x --timeout=600 2>&1 | tail -4
for rep in 1 2; do
  for v in "$A" "$B"; do
    if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
    echo "== $VAR=$v" | tee -a gpurun_out/joints_ab_$TAG.txt
    timeout 600 python tools/bench_joints_batched.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/joints_ab_$TAG.txt
  done
done
