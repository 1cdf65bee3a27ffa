#!/bin/bash
# A/B in one call: fp32 -> (hi, lo) conversion with v_fma_mix (default build) vs the plain C expression (variant library
# lib_split_cform.so, built with -DMORIG_SPLIT_C_FORM for edge_pp.hip and tile_gemm.hip): kernel parity tests on the default build,
# the micro-benchmark shapes, then alternating bench runs.
mkdir -p gpurun_out
OUT=gpurun_out/split_mix_ab.txt
: > $OUT
timeout 1500 python -m pytest tests/test_gpu_kernels.py -q -m gpu --timeout=900 2>&1 | tail -2 | tee -a $OUT
V=$PWD/morig_amd/lib/variants/lib_split_cform.so
for rep in 1 2; do
  MORIG_EDGE_KERNEL= timeout 300 python tools/microbench.py f16x3 16 2>&1 | grep prec= | sed "s/^/mix /" >> $OUT
  MORIG_HIP_LIB=$V timeout 300 python tools/microbench.py f16x3 16 2>&1 | grep prec= | sed "s/^/cform /" >> $OUT
done
grep "^mix\|^cform" $OUT | awk '{k=$1" "$5; if (!(k in mn) || $6<mn[k]) mn[k]=$6} END{for (k in mn) printf "%s  min %.3f ms\n", k, mn[k]}' | sort -k2 | tee -a $OUT.min
for rep in 1 2 3; do
  for v in mix cform; do
    if [ $v = cform ]; then export MORIG_HIP_LIB=$V; else unset MORIG_HIP_LIB; fi
    timeout 300 python bench.py --steps 40 --warmup 8 --secondary 0 --cpu-seconds 0 --prof-steps 0 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench $v', r['value'], r['ms_per_step_median'])" | tee -a $OUT
  done
done
