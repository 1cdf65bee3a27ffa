#!/bin/bash
# Where does the row-local EdgeConv kernel (edge_rl.hip, H = 128) spend its time? Whole-library measurement variants
# (tools/build_variant.sh rl_<V> edge_rl.hip -DRL_<V>; results wrong by construction) against the production build and against the
# producer-consumer kernel (MORIG_RL128=0), alternating rounds in ONE call. noepi = MORIG_DEBUG_FLAGS=1.
TAG=${1:-r05}
mkdir -p gpurun_out
OUT=gpurun_out/edge_rl_ablate_$TAG.txt
: > $OUT
for round in 1 2 3; do
  for v in base pp noepi ${VARIANTS:-NO_RAW NO_CONV NO_DMA NO_W NO_WAIT}; do
    unset MORIG_HIP_LIB MORIG_RL128 MORIG_DEBUG_FLAGS
    case $v in
      base) ;;
      pp) export MORIG_RL128=0 ;;
      noepi) export MORIG_DEBUG_FLAGS=1 ;;
      *) export MORIG_HIP_LIB=$PWD/morig_amd/lib/variants/lib_rl_$v.so ;;
    esac
    MB_NOGEMM=1 MB_HS=128 timeout 120 python tools/microbench.py f16x3 16 2>&1 | grep "edge_" | sed "s/^/$v /" >> $OUT
  done
done
unset MORIG_HIP_LIB MORIG_RL128 MORIG_DEBUG_FLAGS
python - "$OUT" <<'PY'
import sys, collections
d = collections.defaultdict(list)
for ln in open(sys.argv[1]):
    f = ln.split()
    d[(f[0], f[4])].append(float(f[5]))
for s in sorted({k[1] for k in d}):
    print(s, "  ".join(f"{v}={min(d[(v, s)]):.3f}" for v in dict.fromkeys(k[0] for k in d) if (v, s) in d), "ms (min of 3)")
PY
