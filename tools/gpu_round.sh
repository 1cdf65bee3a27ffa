#!/bin/bash
# One validation call: both GPU suites, the edge micro-benchmark A/B, then the default bench line. Outputs in gpurun_out/.
mkdir -p gpurun_out
TAG=${1:-r02a}
timeout 1500 python -m pytest tests/test_gpu_kernels.py -q -m gpu --timeout=900 2>&1 | tail -40 > gpurun_out/pytest_kernels_$TAG.txt
tail -4 gpurun_out/pytest_kernels_$TAG.txt
timeout 1800 python -m pytest tests/test_gpu_networks.py tests/test_gpu_backward.py tests/test_formats.py tests/test_joints_host.py tests/test_harness_and_dist.py -q -m gpu --timeout=1200 2>&1 | tail -60 > gpurun_out/pytest_networks_$TAG.txt
tail -4 gpurun_out/pytest_networks_$TAG.txt
OUT=gpurun_out/ws_ab_$TAG.txt
: > $OUT
for rep in 1 2; do
  for v in ws pp; do MB_NOGEMM=1 MORIG_EDGE_KERNEL=$v timeout 300 python tools/microbench.py f16x3 16 2>&1 | grep prec= | sed "s/^/$v /" >> $OUT; done
done
sort $OUT | awk '{k=$1" "$5; if (!(k in mn) || $6<mn[k]) mn[k]=$6} END{for (k in mn) printf "%s  min %.3f ms\n", k, mn[k]}' | sort -k2
python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
tail -c 2500 gpurun_out/bench_$TAG.json; tail -3 gpurun_out/bench_$TAG.err
