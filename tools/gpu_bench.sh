#!/bin/bash
# bench + rocprofv3 kernel-trace stats on a gpurun box; summaries land in gpurun_out/
mkdir -p gpurun_out
TAG=${1:-r01}
STEPS=${2:-5}
python bench.py --steps $STEPS --warmup ${3:-10} > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
cp gpurun_out/bench_detail.json gpurun_out/bench_detail_$TAG.json 2>/dev/null    # the full record behind the compact last line
tail -n 1 gpurun_out/bench_$TAG.json | head -c 4096; echo; echo "last-line bytes: $(tail -n 1 gpurun_out/bench_$TAG.json | wc -c)"
export TMPDIR=/tmp
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --cpu-seconds 0 --secondary 0 --prof-steps 0 > /tmp/prof_$TAG.log 2>&1 )
f=$(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then head -40 "$f" > gpurun_out/rocprof_kernel_stats_$TAG.csv; else tail -20 /tmp/prof_$TAG.log > gpurun_out/rocprof_$TAG.err; fi
ls gpurun_out
