#!/bin/bash
# The evidence set of a build in ONE call: default bench line, rocprofv3 kernel stats of the same command, the MFMA-utilisation and
# HBM-traffic counter passes (stamped with the library's sha256), the clock / power under load. Everything lands in gpurun_out/.
TAG=${1:-r03}
bash tools/gpu_bench.sh $TAG 50 10 > /dev/null 2>&1
python -c "
import json; r=json.loads(open('gpurun_out/bench_$TAG.json').read().strip().splitlines()[-1]); print('bench', r['value'], r['ms_per_step'], r['roofline']['kernel'], r['roofline']['frac'])"
head -8 gpurun_out/rocprof_kernel_stats_$TAG.csv | cut -c1-160
bash tools/gpu_pmc_mfma.sh $TAG 2>&1 | tail -14
bash tools/gpu_pmc_bench.sh $TAG 2>&1 | tail -12
bash tools/gpu_clocks.sh $TAG 2>&1 | tail -6
