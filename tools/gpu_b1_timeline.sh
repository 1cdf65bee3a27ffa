#!/bin/bash
# timeline of one steady-state EAGER forward at B meshes (default 1): every launch with its grid, in order
mkdir -p gpurun_out
export TMPDIR=/tmp
D=$(pwd); TAG=${1:-b1}; B=${2:-1}
( cd /tmp && rm -rf /tmp/tl_b1 && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_b1 -- python $D/tools/b1_profile.py $B 12 > /tmp/tl_b1.log 2>&1 )
f=$(find /tmp/tl_b1 -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py "$f" cls_attention > gpurun_out/timeline_B${B}_$TAG.txt 2>&1
head -3 gpurun_out/timeline_B${B}_$TAG.txt; tail -1 gpurun_out/timeline_B${B}_$TAG.txt
