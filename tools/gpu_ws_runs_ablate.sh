#!/bin/bash
# What did the run / carry code cost edge_ws<256>? Whole-library variants (tools/build_variant.sh ws_<v> edge_ws.hip -DWS_OLD_SCAN /
# -DWS_OLD_MAP, valid at MORIG_EDGE_RUN=1) against the production build at run 1 and 8, and the previous commit's tree if present
# (_old/), sustained (tools/op_clock.py, OP_SPLIT=1: split-fp16 rows as in the networks), two rounds in ONE call.
mkdir -p gpurun_out
OUT=gpurun_out/ws_runs_ablate_${1:-r05}.txt; : > $OUT
for round in 1 2; do
  for v in ${VS:-run8 run1 prev}; do
    unset MORIG_HIP_LIB; export MORIG_EDGE_RUN=1 OP_SPLIT=1; d=.
    case $v in
      run8) export MORIG_EDGE_RUN=8 ;;
      run1) ;;
      run8f) export MORIG_EDGE_RUN=8; unset OP_SPLIT ;;        # fp32 rows out (no conversion pass, the <.., false> kernels)
      run1f) unset OP_SPLIT ;;
      prev) d=_old; unset OP_SPLIT; [ -d _old ] || continue ;;
      r8_*) export MORIG_EDGE_RUN=8 MORIG_HIP_LIB=$PWD/morig_amd/lib/variants/lib_ws_${v#r8_}.so ;;      # a variant library at run 8
      *) export MORIG_HIP_LIB=$PWD/morig_amd/lib/variants/lib_ws_$v.so ;;
    esac
    ( cd $d && OP_ONLY=${OPK:-edge_geo_H256} timeout 200 python tools/op_clock.py 2 2>&1 | grep edge_geo | sed "s/^/$v /" ) | tee -a $OUT
  done
done
