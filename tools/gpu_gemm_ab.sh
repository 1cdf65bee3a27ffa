#!/bin/bash
# DMA GEMM variants in one call: parity tests on the default build, then the micro-benchmark shapes per library variant
mkdir -p gpurun_out
TAG=${1:-a}
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "gemm" --timeout=600 2>&1 | tail -8 > gpurun_out/gemm_tests_$TAG.txt
tail -3 gpurun_out/gemm_tests_$TAG.txt
OUT=gpurun_out/gemm_ab_$TAG.txt
: > $OUT
run() { label=$1; shift; env "$@" MB_NOEDGE=1 timeout 300 python tools/microbench.py f16x3 16 2>&1 | grep -E "gemm16" | sed "s/^/$label /" >> $OUT; }
for rep in 1 2 3; do
  run new X=1
  for f in morig_amd/lib/variants/lib_*.so; do v=$(basename $f .so); v=${v#lib_}; [ $v = trace ] && continue; run $v MORIG_HIP_LIB=$PWD/$f; done
done
sort $OUT | awk '{k=$1" "$5; if (!(k in mn) || $6<mn[k]) mn[k]=$6} END{for (k in mn) printf "%s  min %.3f ms\n", k, mn[k]}' | sort -k2
