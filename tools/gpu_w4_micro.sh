#!/bin/bash
# four-wave EdgeConv quick loop: parity test, then the H = 256 micro-benchmark with and without the epilogue (MORIG_DEBUG_FLAGS=1) for
# the eight-wave kernel, the four-wave kernel and variant libraries. usage: tools/gpu_w4_micro.sh <tag> [variant ...]
TAG=${1:-w4m}; shift
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x --timeout=240 -k "four_wave" 2>&1 | tail -3
OUT=gpurun_out/w4_micro_$TAG.txt; : > $OUT
for rep in 1 2 3; do for dbg in 0 1; do
  MORIG_DEBUG_FLAGS=$dbg MORIG_EDGE_W4=0 MB_NOGEMM=1 MB_HS=256 timeout 300 python tools/microbench.py f16x3 16 2>&1 | grep edge_ | sed "s/^/ws8 /" >> $OUT
  MORIG_DEBUG_FLAGS=$dbg MORIG_EDGE_W4=1 MB_NOGEMM=1 MB_HS=256 timeout 300 python tools/microbench.py f16x3 16 2>&1 | grep edge_ | sed "s/^/w4 /" >> $OUT
  for v in "$@"; do
    MORIG_HIP_LIB=$PWD/morig_amd/lib/variants/lib_$v.so MORIG_DEBUG_FLAGS=$dbg MORIG_EDGE_W4=1 MB_NOGEMM=1 MB_HS=256 timeout 300 python tools/microbench.py f16x3 16 2>&1 | grep edge_ | sed "s/^/$v /" >> $OUT
  done
done; done
awk '{k=$1" dbg"$4" "$5; v=$6; if (!(k in mn) || v<mn[k]) mn[k]=v} END{for (k in mn) printf "%s  min %.3f ms\n", k, mn[k]}' $OUT | sort -k3 -k2 | tee gpurun_out/w4_micro_summary_$TAG.txt
