#!/bin/bash
# tools/microbench.py under a list of MORIG_DEBUG_FLAGS settings (phase ablations) -> gpurun_out/microbench_<tag>.txt
mkdir -p gpurun_out
OUT=gpurun_out/microbench_${1:-x}.txt
FLAGS=${2:-"0 1"}
: > $OUT
for f in $FLAGS ; do MORIG_DEBUG_FLAGS=$f python tools/microbench.py f16x3 16 >> $OUT 2>&1; done
if [ -n "$3" ]; then for f in $FLAGS ; do env $3 MORIG_DEBUG_FLAGS=$f python tools/microbench.py f16x3 16 | sed "s/^/[$3] /" >> $OUT 2>&1; done; fi
grep -v amdgpu.ids $OUT
