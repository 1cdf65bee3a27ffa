#!/bin/bash
# timeline of one steady-state step of a bench workload: tools/gpu_timeline.sh <tag> <workload> <marker kernel>
mkdir -p gpurun_out
export TMPDIR=/tmp
D=$(pwd); TAG=$1; W=$2; M=$3
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$W -- python $D/bench.py --workload $W --steps 6 --warmup 3 --cpu-seconds 0 --secondary 0 --prof-steps 0 > /tmp/tl_$W.log 2>&1 )
f=$(find /tmp/tl_$W -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py "$f" "$M" > gpurun_out/timeline_${W}_$TAG.txt 2>&1
head -5 gpurun_out/timeline_${W}_$TAG.txt
