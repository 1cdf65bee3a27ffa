#!/bin/bash
mkdir -p gpurun_out
MORIG_HIP_LIB=$PWD/morig_amd/lib/variants/lib_trace.so python tools/mix_micro.py 0.02 > gpurun_out/ws_trace_$1.txt 2>&1
grep -A8 "WS_TRACE" gpurun_out/ws_trace_$1.txt | awk 'BEGIN{n=0} /WS_TRACE/{n++} {print}' | head -400 > gpurun_out/ws_trace_$1.short
grep -n "^tpl\|^geo" gpurun_out/ws_trace_$1.txt
