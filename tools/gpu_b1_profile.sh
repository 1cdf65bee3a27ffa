#!/bin/bash
# per-kernel time of the B = 1 forward (rocprofv3 kernel stats over 40 eager forwards)
TAG=${1:-b1}; B=${2:-1}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/b1prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/b1prof -- python $GRAFT_REPO_ROOT/tools/b1_profile.py $B 40 > /tmp/b1prof.log 2>&1
f=$(find /tmp/b1prof -name "*kernel_stats.csv" | head -1)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
if [ -n "$f" ]; then head -60 "$f" > $GRAFT_REPO_ROOT/gpurun_out/rocprof_kernel_stats_B${B}_$TAG.csv; cut -c1-110 "$f" | head -45; else tail -20 /tmp/b1prof.log; fi
