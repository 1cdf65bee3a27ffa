#!/bin/bash
# round 6: whole GPU suite + the default bench line (ragged / served small batches in `secondary`)
mkdir -p gpurun_out
TAG=${1:-r06b}
timeout 1500 python -m pytest tests -q -m gpu -x --timeout=900 2>&1 | tail -70 > gpurun_out/pytest_gpu_$TAG.txt; tail -5 gpurun_out/pytest_gpu_$TAG.txt
python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; tail -n 1 gpurun_out/bench_$TAG.json | head -c 4096; echo
cp gpurun_out/bench_detail.json gpurun_out/bench_detail_$TAG.json
