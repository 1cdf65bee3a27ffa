#!/bin/bash
# full validation of a build: every -m gpu test (as the driver runs them), smoke, the default bench line
mkdir -p gpurun_out
TAG=${1:-full}
timeout 2400 python -m pytest tests -q -m gpu --timeout=1200 2>&1 | tail -45 > gpurun_out/pytest_gpu_$TAG.txt; tail -45 gpurun_out/pytest_gpu_$TAG.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
cp gpurun_out/bench_detail.json gpurun_out/bench_detail_$TAG.json 2>/dev/null
python -c "
import json; r=json.loads(open('gpurun_out/bench_$TAG.json').read().strip().splitlines()[-1])
print(r['value'], r['ms_per_step'], r['ms_per_step_median']); s=r['secondary']
print({k: (v.get('value'), v.get('ms_per_step')) for k, v in s.items() if isinstance(v, dict) and 'value' in v}); print(json.dumps(s['small_batch'])); print(json.dumps(s['joint_extraction'])); print('last-line bytes', len(open('gpurun_out/bench_$TAG.json').read().strip().splitlines()[-1]))"; tail -3 gpurun_out/bench_$TAG.err
