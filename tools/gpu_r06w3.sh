#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r06w_gpu_auto.txt 2>&1; echo "suite auto: $?" > gpurun_out/r06w_rc.txt
timeout 600 python - > gpurun_out/r06w_odd_off.txt 2>&1 <<'PY'
# what the two odd-magnitude layers did BEFORE the row normalisation: the same test body with the factors forced to 1
import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_gpu_networks as t
import morig_amd.packing as P
P._row_factors = lambda W: torch.ones(W.shape[0], device=W.device)
try:
    t.test_row_normalised_weights_keep_odd_magnitudes_accurate_and_on_the_fast_path()
    print("OFF: passed (unexpected)")
except AssertionError as e:
    print("OFF: fails as expected:", str(e)[:300])
PY
cat gpurun_out/r06w_rc.txt; tail -3 gpurun_out/r06w_odd_off.txt
grep -n "passed\|failed\|^FAILED" gpurun_out/r06w_gpu_auto.txt | tail
