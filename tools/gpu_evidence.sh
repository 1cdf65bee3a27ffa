#!/bin/bash
# The evidence set of a build in ONE call: default bench line, rocprofv3 kernel stats of the same command, the MFMA-utilisation and
# HBM-traffic counter passes (stamped with the library's sha256), the clock / power under load. Everything lands in gpurun_out/.
TAG=${1:-r03}
bash tools/gpu_bench.sh $TAG 50 10 > /dev/null 2>&1
python -c "
import json; r=json.loads(open('gpurun_out/bench_$TAG.json').read().strip().splitlines()[-1]); print('bench', r['value'], r['ms_per_step'], r['roofline']['kernel'], r['roofline']['frac'])"
head -8 gpurun_out/rocprof_kernel_stats_$TAG.csv | cut -c1-160
# BASELINE configs[2] / configs[3]: rocprofv3 kernel stats of their own bench commands (VERDICT r3 #1a)
export TMPDIR=/tmp
for wl in mask_skin corrnet; do
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${TAG}_$wl -- python $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 5 --warmup 2 --cpu-seconds 0 --secondary 0 --prof-steps 0 > /tmp/prof_${TAG}_$wl.log 2>&1 )
  f=$(find /tmp/prof_${TAG}_$wl -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then head -40 "$f" > gpurun_out/rocprof_kernel_stats_${wl}_$TAG.csv; head -4 "$f" | cut -c1-140; else tail -5 /tmp/prof_${TAG}_$wl.log; fi
done
bash tools/gpu_pmc_mfma.sh $TAG 2>&1 | tail -14
bash tools/gpu_pmc_bench.sh $TAG 2>&1 | tail -12
bash tools/gpu_clocks.sh $TAG 2>&1 | tail -6
