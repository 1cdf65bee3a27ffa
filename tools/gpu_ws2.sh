#!/bin/bash
# edge_ws.hip ablations in one call: library variants built with -DWS_NO_* (morig_amd/lib/variants) + debug flags
mkdir -p gpurun_out
TAG=${1:-b}
OUT=gpurun_out/ws_ablate_$TAG.txt
: > $OUT
run() { label=$1; shift; env "$@" MB_NOGEMM=1 MB_HS=${HS:-256,128} timeout 300 python tools/microbench.py f16x3 16 2>&1 | grep -E "prec=|WS_TRACE" | sed "s/^/$label /" >> $OUT; }
run full X=1
run full X=1
run noepi MORIG_DEBUG_FLAGS=1
for v in noconv nodma nofrag nobar; do run $v MORIG_HIP_LIB=$PWD/morig_amd/lib/variants/lib_$v.so; done
run trace MORIG_HIP_LIB=$PWD/morig_amd/lib/variants/lib_trace.so
run pp MORIG_EDGE_KERNEL=pp
cat $OUT
