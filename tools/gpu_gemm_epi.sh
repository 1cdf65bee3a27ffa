#!/bin/bash
# What is the store epilogue of the LDS-DMA GEMM made of? Whole-library measurement variants (tools/build_variant.sh, results wrong
# by construction): no global stores / no LDS transposition writes / no LDS at all / neither, against the full kernel and the
# no-epilogue build (MORIG_DEBUG_FLAGS=1), all in one call.
mkdir -p gpurun_out
TAG=${1:-a}
OUT=gpurun_out/gemm_epi_$TAG.txt
: > $OUT
run() { label=$1; shift; env "$@" MORIG_DMA_PERSIST=0 MB_NOEDGE=1 timeout 300 python tools/microbench.py f16x3 16 2>&1 | grep -E "gemm16" | sed "s/^/$label /" >> $OUT; }
for rep in 1 2; do
  run full X=1
  run noepi MORIG_DEBUG_FLAGS=1
  for f in morig_amd/lib/variants/lib_epi_*.so; do v=$(basename $f .so); v=${v#lib_}; run $v MORIG_HIP_LIB=$PWD/$f; done
done
sort $OUT | awk '{k=$1" "$5; if (!(k in mn) || $6<mn[k]) mn[k]=$6} END{for (k in mn) printf "%-18s %s  min %.3f ms\n", $0="", k, mn[k]}' | sort -k2,2 -k4n
