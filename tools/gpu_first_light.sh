#!/bin/bash
# First-light script for a gpurun box: environment facts, then the GPU suites without -x so one call
# reports every failing kernel. Output lands in gpurun_out/ (merged back by gpurun).
mkdir -p gpurun_out
{
  echo "== rocminfo =="; /opt/rocm/bin/rocminfo | grep -E "Marketing Name|gfx|Compute Unit" | head -8
  echo "== lscpu =="; lscpu | grep -E "Model name|^CPU\(s\)|Thread|Core|Socket" 
  python - <<'PY'
from morig_amd import native
print("device_info:", native.device_info())
PY
} > gpurun_out/env.txt 2>&1
python -m pytest tests/test_gpu_kernels.py -q -m gpu --timeout=600 2>&1 | tail -60 > gpurun_out/pytest_kernels.txt
python -m pytest tests/test_gpu_networks.py -q -m gpu --timeout=900 2>&1 | tail -80 > gpurun_out/pytest_networks.txt
tail -5 gpurun_out/pytest_kernels.txt; tail -5 gpurun_out/pytest_networks.txt
