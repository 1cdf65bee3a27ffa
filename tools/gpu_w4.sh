#!/bin/bash
# four-wave EdgeConv (edge_w4.hip): its parity test alone under a short timeout (a schedule bug could hang), then the EdgeConv kernel
# tests and the network goldens with MORIG_EDGE_W4=1, the micro-benchmark pair, and the A/B of the switch (alternating bench runs).
# usage: tools/gpu_w4.sh <tag>
TAG=${1:-w4}
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x --timeout=240 -k "four_wave" 2>&1 | tail -15 > gpurun_out/pytest_w4_$TAG.txt; tail -8 gpurun_out/pytest_w4_$TAG.txt
if ! grep -q " passed" gpurun_out/pytest_w4_$TAG.txt || grep -q "failed\|error" gpurun_out/pytest_w4_$TAG.txt; then echo "four-wave parity test did not pass: stopping"; exit 1; fi
for rep in 1 2 3; do for v in 0 1; do
  MORIG_EDGE_W4=$v MB_NOGEMM=1 MB_HS=256 timeout 300 python tools/microbench.py f16x3 16 2>&1 | grep edge_ | sed "s/^/w4=$v /"
done; done | tee gpurun_out/w4_micro_$TAG.txt
MORIG_EDGE_W4=1 timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x --timeout=600 -k "edgeconv" 2>&1 | tail -4
MORIG_EDGE_W4=1 timeout 1200 python -m pytest tests/test_gpu_networks.py -q -m gpu -x --timeout=900 2>&1 | tail -3
bash tools/gpu_env_ab.sh $TAG MORIG_EDGE_W4 0 1 3
