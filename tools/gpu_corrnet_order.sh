#!/bin/bash
# CorrNet enqueue order A/B in one call (MORIG_CORRNET_ORDER=vertex_first or points_first): parity tests, then alternating bench runs
mkdir -p gpurun_out
OUT=gpurun_out/corrnet_order_ab.txt
: > $OUT
timeout 900 python -m pytest tests/test_gpu_networks.py -q -m gpu -k "corrnet or deformnet or point" --timeout=600 2>&1 | tail -2 | tee -a $OUT
for rep in 1 2 3; do
  for v in points_first vertex_first; do
    for w in corrnet deformnet; do
      MORIG_CORRNET_ORDER=$v timeout 300 python bench.py --workload $w --steps 60 --warmup 10 --secondary 0 --cpu-seconds 0 --prof-steps 0 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w order=$v', r['value'], r['ms_per_step_median'])" | tee -a $OUT
    done
  done
done
