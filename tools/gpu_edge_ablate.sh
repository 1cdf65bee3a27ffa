#!/bin/bash
# edge_ws.hip ablations / A-B in ONE gpurun call (box-to-box variance is ~3 %): parity tests on the default build, then the
# micro-benchmark per library variant. Variants are whole libraries built with -DWS_NO_CONVERT / -DWS_NO_DMA / -DWS_NO_FRAG /
# -DWS_NO_BARRIER / -DMORIG_WS_TRACE ... into morig_amd/lib/variants/lib_<name>.so (hipcc ... -c edge_ws.hip, relink with the
# other objects) and selected with MORIG_HIP_LIB. Results of round 2: profiles/r02_edge_ws_ablations_*.txt
mkdir -p gpurun_out
TAG=${1:-c}
MORIG_WS128=1 timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "edgeconv" --timeout=600 2>&1 | tail -12 > gpurun_out/ws_tests_$TAG.txt
tail -3 gpurun_out/ws_tests_$TAG.txt
OUT=gpurun_out/ws_ablate_$TAG.txt
: > $OUT
run() { label=$1; shift; env "$@" MORIG_WS128=1 MB_NOGEMM=1 MB_HS=${HS:-256,128} timeout 300 python tools/microbench.py f16x3 16 2>&1 | grep -E "prec=|WS_TRACE" | sed "s/^/$label /" >> $OUT; }
for rep in 1 2 3; do
  run full X=1
  for f in morig_amd/lib/variants/lib_*.so; do v=$(basename $f .so); v=${v#lib_}; [ $v = trace ] && [ $rep != 1 ] && continue; run $v MORIG_HIP_LIB=$PWD/$f; done
  run pp MORIG_EDGE_KERNEL=pp
done
run noepi MORIG_DEBUG_FLAGS=1
grep -v WS_TRACE $OUT | sort | awk '{k=$1" "$5; if (!(k in mn) || $6<mn[k]) mn[k]=$6; n[k]++} END{for (k in mn) printf "%s  min %.3f ms (n=%d)\n", k, mn[k], n[k]}' | sort -k2
grep WS_TRACE $OUT | head -8
