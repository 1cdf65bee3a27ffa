#!/bin/bash
# secondary configs (BASELINE.json configs[2], configs[3]) on one GPU
mkdir -p gpurun_out
TAG=${1:-r01}
python bench.py --workload mask_skin --steps 3 --warmup 1 > gpurun_out/bench_maskskin_$TAG.json 2> gpurun_out/bench_maskskin_$TAG.err
tail -c 2500 gpurun_out/bench_maskskin_$TAG.json
python bench.py --workload corrnet --steps 3 --warmup 1 > gpurun_out/bench_corrnet_$TAG.json 2> gpurun_out/bench_corrnet_$TAG.err
tail -c 2500 gpurun_out/bench_corrnet_$TAG.json; tail -3 gpurun_out/bench_corrnet_$TAG.err
python bench.py --workload deformnet --steps 3 --warmup 1 --cpu-seconds 0 > gpurun_out/bench_deformnet_$TAG.json 2> gpurun_out/bench_deformnet_$TAG.err
tail -c 1500 gpurun_out/bench_deformnet_$TAG.json; tail -3 gpurun_out/bench_deformnet_$TAG.err
