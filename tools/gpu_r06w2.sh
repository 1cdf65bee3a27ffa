#!/bin/bash
cd /root/repo; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r06w_gpu_auto.txt 2>&1; echo "suite auto: $?" > gpurun_out/r06w_rc.txt
MORIG_PACK_NORMALISE=0 timeout 600 python - > gpurun_out/r06w_odd_off.txt 2>&1 <<'PY'
import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_gpu_networks as t
from morig_amd import packing
packing.PACK_NORMALISE = "auto"      # let the test body run ...
import morig_amd.packing as P
# ... but with the normalisation itself disabled, to record what the layers did before
P._row_factors_orig = P._row_factors
P._row_factors = lambda W: torch.ones(W.shape[0])
try:
    t.test_row_normalised_weights_keep_odd_magnitudes_accurate_and_on_the_fast_path()
    print("OFF: passed (unexpected)")
except AssertionError as e:
    print("OFF: fails as expected:", str(e)[:300])
PY
for n in 0 auto 0 auto; do
  MORIG_PACK_NORMALISE=$n python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r06w_b.json
  python - <<PY >> gpurun_out/r06w_rc.txt
import json; d=json.load(open("gpurun_out/r06w_b.json")); print("normalise=$n", d["value"], d["ms_per_step"], d["roofline"].get("frac"))
PY
done
cat gpurun_out/r06w_rc.txt; cat gpurun_out/r06w_odd_off.txt | tail -3
grep -n "passed\|failed\|^FAILED" gpurun_out/r06w_gpu_auto.txt | tail
