#!/bin/bash
# DMA GEMM A/B in one call: parity tests, then the micro-benchmark shapes with and without the persistent kernel (gemm_dmap.hip)
mkdir -p gpurun_out
TAG=${1:-a}
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "gemm" --timeout=600 2>&1 | tail -6 > gpurun_out/gemm_tests_$TAG.txt
tail -3 gpurun_out/gemm_tests_$TAG.txt
OUT=gpurun_out/gemm_ab_$TAG.txt
: > $OUT
for rep in 1 2 3; do
  for v in 1 0; do MORIG_DMA_PERSIST=$v MB_NOEDGE=1 timeout 300 python tools/microbench.py f16x3 16 2>&1 | grep -E "gemm16" | sed "s/^/persist$v /" >> $OUT; done
done
sort $OUT | awk '{k=$1" "$5; if (!(k in mn) || $6<mn[k]) mn[k]=$6} END{for (k in mn) printf "%s  min %.3f ms\n", k, mn[k]}' | sort -k2
for v in 1 0; do MORIG_DMA_PERSIST=$v python bench.py --secondary 0 --cpu-seconds 0 --steps 30 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('persist$v', d['value'], d['ms_per_step'], {k: v['ms_per_step'] for k, v in list(d['kernels'].items())[:3]})"; done
