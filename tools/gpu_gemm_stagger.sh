#!/bin/bash
# Staggered start of the first workgroup per CU (MORIG_DMA_STAGGER = number of phases) against the lock-step launch, one call.
mkdir -p gpurun_out
TAG=${1:-a}
OUT=gpurun_out/gemm_stagger_$TAG.txt
: > $OUT
for rep in 1 2; do
  for ph in 0 2 4 8; do
    MORIG_DMA_STAGGER=$ph MORIG_DMA_PERSIST=0 MB_NOEDGE=1 timeout 300 python tools/microbench.py f16x3 16 2>&1 | grep -E "gemm16" | sed "s/^/phases$ph /" >> $OUT
  done
done
sort $OUT | awk '{k=$1" "$5; if (!(k in mn) || $6<mn[k]) mn[k]=$6} END{for (k in mn) printf "%s  min %.3f ms\n", k, mn[k]}' | sort -k2
