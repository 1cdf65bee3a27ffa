#!/bin/bash
# eight-wave EdgeConv (edge_ws.hip) variants on the H = 256 / H = 128 micro-benchmark, alternating in ONE call; parity tests first.
# usage: tools/gpu_ws_micro.sh <tag> [variant ...]   (a variant = morig_amd/lib/variants/lib_<name>.so; "default" = the regular build)
TAG=${1:-wsm}; shift
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x --timeout=500 -k "edgeconv" 2>&1 | tail -2
OUT=gpurun_out/ws_micro_$TAG.txt; : > $OUT
for rep in 1 2 3 4; do
  MORIG_EDGE_W4=0 MB_NOGEMM=1 MB_HS=256 timeout 300 python tools/microbench.py f16x3 16 2>&1 | grep edge_ | sed "s/^/default /" >> $OUT
  for v in "$@"; do
    MORIG_HIP_LIB=$PWD/morig_amd/lib/variants/lib_$v.so MORIG_EDGE_W4=0 MB_NOGEMM=1 MB_HS=256 timeout 300 python tools/microbench.py f16x3 16 2>&1 | grep edge_ | sed "s/^/$v /" >> $OUT
  done
done
awk '{k=$1" "$5; v=$6; if (!(k in mn) || v<mn[k]) mn[k]=v; s[k]+=v; n[k]++} END{for (k in mn) printf "%s  min %.3f  mean %.3f ms\n", k, mn[k], s[k]/n[k]}' $OUT | sort -k2 | tee gpurun_out/ws_micro_summary_$TAG.txt
