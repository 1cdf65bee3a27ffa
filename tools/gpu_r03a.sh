#!/bin/bash
# r03 first call: kernel-suite sanity on the new build, the small-batch sweep, the hipGraph replay bisect, then the default bench line.
mkdir -p gpurun_out
TAG=${1:-r03a}
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x --timeout=600 2>&1 | tail -5 > gpurun_out/pytest_kernels_$TAG.txt
tail -3 gpurun_out/pytest_kernels_$TAG.txt
timeout 600 python tools/small_batch.py > gpurun_out/small_batch_$TAG.txt 2>&1; tail -30 gpurun_out/small_batch_$TAG.txt
for nb in 2 16; do timeout 600 python tools/graph_bisect.py $nb 4 > gpurun_out/graph_bisect_${TAG}_b$nb.txt 2>&1; tail -25 gpurun_out/graph_bisect_${TAG}_b$nb.txt; done
python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
tail -c 3000 gpurun_out/bench_$TAG.json; tail -3 gpurun_out/bench_$TAG.err
