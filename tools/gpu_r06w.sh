#!/bin/bash
# row normalisation of the packed weights (packing.py): parity of the full GPU suite with it on, the network-level error summary with it
# off / on, and the bench line off / on / off / on (same box)
cd /root/repo; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r06w_gpu_on.txt 2>&1; echo "suite on: $?" > gpurun_out/r06w_rc.txt
MORIG_PACK_NORMALISE=0 timeout 1200 python -m pytest tests/test_gpu_networks.py -m gpu -q > gpurun_out/r06w_net_off.txt 2>&1; echo "net off: $?" >> gpurun_out/r06w_rc.txt
for i in 1 2; do
  for n in 0 1; do
    MORIG_PACK_NORMALISE=$n python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r06w_bench_${n}_$i.json
    python - <<PY >> gpurun_out/r06w_rc.txt
import json; d=json.load(open("gpurun_out/r06w_bench_${n}_$i.json")); print("normalise=$n", d["value"], d["ms_per_step"], d["roofline"].get("frac"))
PY
  done
done
cat gpurun_out/r06w_rc.txt
tail -5 gpurun_out/r06w_gpu_on.txt
