#!/bin/bash
# A/B two builds of the library inside ONE gpurun call (box-to-box variance is ~5 %): tools/ab_edge.sh libA.so libB.so [pattern]
A=$1; B=$2; PAT=${3:-edge}
for i in 1 2 3; do
  for v in A B; do
    if [ $v = A ]; then L=$A; else L=$B; fi
    MORIG_HIP_LIB=$PWD/$L timeout 200 python tools/microbench.py f16x3 16 2>&1 | grep -E "$PAT" | sed "s/^/$v /"
  done
done | sort | awk '{k=$1" "$5; t[k]+=$6; n[k]++} END{for (k in t) printf "%s  %.3f ms (n=%d)\n", k, t[k]/n[k], n[k]}' | sort -k2
