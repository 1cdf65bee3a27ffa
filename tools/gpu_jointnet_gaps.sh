#!/bin/bash
# idle time of the GPU inside one steady-state jointnet forward (kernel trace of the bench command)
mkdir -p gpurun_out
export TMPDIR=/tmp
D=$(pwd)
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/gap_j -- python $D/bench.py --steps 6 --warmup 3 --cpu-seconds 0 --secondary 0 --prof-steps 0 > /tmp/gap_j.log 2>&1 )
f=$(find /tmp/gap_j -name "*kernel_trace.csv" | head -1)
python tools/gap_report.py "$f" 10 cls_attention_kernel 2>&1 | head -40 | tee gpurun_out/jointnet_gaps.txt
