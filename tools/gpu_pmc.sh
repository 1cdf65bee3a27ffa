#!/bin/bash
# PMC counters for the hot kernels (own passes, kernel-trace only, per MI355X_MICROARCH.md)
mkdir -p gpurun_out
TAG=${1:-x}
export TMPDIR=/tmp
cd /tmp
run() {  # name, counters...
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$name -- python $GRAFT_REPO_ROOT/tools/microbench.py f16x3 8 > /tmp/pmc_$name.log 2>&1
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python - "$f" "$name" <<'PY' >> $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}.txt
import csv, sys, collections
f, name = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for row in csv.DictReader(open(f)):
    k = row["Kernel_Name"]
    if "tile_kernel" not in k: continue
    k = k.split("(")[0].replace("void morig::", "")
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); 
    cnt[(k, row["Counter_Name"])] += 1
for k in acc:
    print(name, k, {c: round(v / cnt[(k, c)], 1) for c, v in acc[k].items()}, "dispatches", max(cnt[(k, c)] for c in acc[k]))
PY
  else tail -5 /tmp/pmc_$name.log >> $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}.txt; fi
}
: > $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}.txt
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_MFMA
run sq3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC
run grbm GRBM_GUI_ACTIVE
cat $GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}.txt
