#!/bin/bash
# ping-pong GEMM variants against the in-phase persistent kernel on the micro-benchmark's deep-wide shapes, alternating in ONE call,
# then one bench pair per variant. usage: tools/gpu_pp_micro.sh <tag> [variant libs ...]   (a variant = morig_amd/lib/variants/lib_<name>.so)
TAG=${1:-ppm}; shift
mkdir -p gpurun_out
OUT=gpurun_out/pp_micro_$TAG.txt; : > $OUT
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x --timeout=240 -k "ping_pong" 2>&1 | tail -3
for rep in 1 2 3; do
  MORIG_GEMM_PP=0 MB_NOEDGE=1 timeout 300 python tools/microbench.py f16x3 32 2>&1 | grep -E "gemm16" | sed "s/^/inphase /" >> $OUT
  MORIG_GEMM_PP=1 MB_NOEDGE=1 timeout 300 python tools/microbench.py f16x3 32 2>&1 | grep -E "gemm16" | sed "s/^/pp_default /" >> $OUT
  for v in "$@"; do
    MORIG_HIP_LIB=$PWD/morig_amd/lib/variants/lib_$v.so MORIG_GEMM_PP=1 MB_NOEDGE=1 timeout 300 python tools/microbench.py f16x3 32 2>&1 | grep -E "gemm16" | sed "s/^/$v /" >> $OUT
  done
done
awk '{k=$1" "$5; v=$6; if (!(k in mn) || v<mn[k]) mn[k]=v} END{for (k in mn) printf "%s  min %.3f ms\n", k, mn[k]}' $OUT | sort -k2 | grep "K1862\|K832" | tee gpurun_out/pp_micro_summary_$TAG.txt
for v in 0 1; do MORIG_GEMM_PP=$v python bench.py --secondary 0 --cpu-seconds 0 --steps 30 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('MORIG_GEMM_PP=$v', d['value'], d['ms_per_step_median'], {k: v['ms_per_step'] for k, v in list(d['kernels'].items())[:4]})" | tee -a gpurun_out/pp_micro_summary_$TAG.txt; done
