#!/bin/bash
# edge_ws.hip bring-up: kernel parity tests, then A/B against the producer/consumer kernel in ONE call
mkdir -p gpurun_out
TAG=${1:-a}
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "edgeconv" --timeout=600 -x 2>&1 | tail -15 > gpurun_out/ws_tests_$TAG.txt
tail -5 gpurun_out/ws_tests_$TAG.txt
: > gpurun_out/ws_bench_$TAG.txt
for i in 1 2; do
  for v in ws pp; do
    MB_NOGEMM=1 MORIG_EDGE_KERNEL=$v timeout 300 python tools/microbench.py f16x3 16 2>&1 | grep prec= | sed "s/^/$v /" >> gpurun_out/ws_bench_$TAG.txt
  done
done
cat gpurun_out/ws_bench_$TAG.txt
