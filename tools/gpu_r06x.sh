#!/bin/bash
# fork / join of the tpl and geo EdgeConv launches: thresholds 0 (off) / 1e9 (always), interleaved twice
cd /root/repo; mkdir -p gpurun_out; : > gpurun_out/r06x_fork.txt
for i in 1 2; do for r in 0 1000000000; do
  MORIG_FORK_ROWS=$r timeout 600 python tools/fork_ab.py 2>&1 | tail -1 >> gpurun_out/r06x_fork.txt
done; done
cat gpurun_out/r06x_fork.txt
