#!/bin/bash
# idle time of the GPU inside one steady-state CorrNet / DeformNet forward (kernel trace; concurrent streams counted once)
mkdir -p gpurun_out
export TMPDIR=/tmp
D=$(pwd)
for w in corrnet; do
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/gap_$w -- python $D/bench.py --workload $w --steps 6 --warmup 3 --cpu-seconds 0 --secondary 0 --prof-steps 0 > /tmp/gap_$w.log 2>&1 )
  f=$(find /tmp/gap_$w -name "*kernel_trace.csv" | head -1)
  echo "== $w" | tee -a gpurun_out/corrnet_gaps.txt
  python tools/gap_report.py "$f" 15 make_seg_kernel 2>&1 | head -34 | tee -a gpurun_out/corrnet_gaps.txt
done
