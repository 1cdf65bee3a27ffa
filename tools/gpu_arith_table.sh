#!/bin/bash
# DESIGN section 3 "per-layer arithmetic" measurement: end-to-end error of the network goldens (both BN recipes, full sizes) with the
# 2-MFMA variants (W rounded to fp16 in the LDS-DMA GEMMs / in the H=256 EdgeConv second layers / both) against the 3-MFMA default.
mkdir -p gpurun_out
TAG=${1:-a}
OUT=gpurun_out/arith_table_$TAG.txt
: > $OUT
for v in default 2mfma_gemm 2mfma_edge 2mfma_both; do
  echo "=== $v" >> $OUT
  if [ $v = default ]; then L=""; else L="MORIG_HIP_LIB=$PWD/morig_amd/lib/variants/lib_$v.so"; fi
  env $L timeout 900 python -m pytest tests/test_gpu_networks.py -q -m gpu -k "full_size or headline_size or reference_goldens or corrnet_against" --timeout=600 2>&1 | grep -E "^ [0-9]\.[0-9]+e|passed|failed" >> $OUT
  env $L MB_NOEDGE=1 timeout 300 python tools/microbench.py f16x3 16 2>&1 | grep gemm16 >> $OUT
  env $L MB_NOGEMM=1 MB_HS=256 timeout 300 python tools/microbench.py f16x3 16 2>&1 | grep prec= >> $OUT
done
cat $OUT
