#!/bin/bash
# Where does the persistent LDS-DMA GEMM (gemm_dmap.hip) lose its 35 % of MFMA issue? Whole-library measurement variants
# (tools/build_variant.sh dmap_<V> gemm_dmap.hip -DDMAP_ABL_<V>; results wrong by construction) against the production build, three
# alternating rounds in ONE call: HOT = every LDS-DMA re-reads the same 64 KB (the instructions without the memory behind them),
# NODMA = no LDS-DMA in the steady state, NOWAIT = nobody waits for a chunk to land, NOFRAG = no fragment reads in the steady state.
TAG=${1:-r05}
mkdir -p gpurun_out
OUT=gpurun_out/gemm_dmap_ablate_$TAG.txt
: > $OUT
for round in 1 2 3; do
  for v in base HOT NODMA NOWAIT NOFRAG ${EXTRA_VARIANTS}; do
    if [ $v = base ]; then unset MORIG_HIP_LIB; else export MORIG_HIP_LIB=$PWD/morig_amd/lib/variants/lib_dmap_$v.so; fi
    MB_NOEDGE=1 python tools/microbench.py f16x3 16 2>&1 | grep "gemm16_" | sed "s/^/$v /" >> $OUT
  done
done
unset MORIG_HIP_LIB
python - "$OUT" <<'PY'
import sys, collections
d = collections.defaultdict(list)
for ln in open(sys.argv[1]):
    f = ln.split()
    d[(f[0], f[3])].append(float(f[4]))
shapes = sorted({k[1] for k in d})
for s in shapes:
    print(s, "  ".join(f"{v}={min(d[(v, s)]):.3f}" for v in dict.fromkeys(k[0] for k in d) if (v, s) in d), "ms (min of 3)")
PY
