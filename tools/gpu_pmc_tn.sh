#!/bin/bash
# counters of the weight-gradient GEMM (gemm_tn16_kernel) on the training step's shapes: MFMA busy, VALU / LDS / VMEM activity, LDS bank
# conflicts, HBM fetch -- each group in its own pass, kernel-trace only. usage: tools/gpu_pmc_tn.sh <tag>
mkdir -p gpurun_out
TAG=${1:-x}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_tn_${TAG}.txt
: > $OUT
cd /tmp
run() {
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$name -- python $GRAFT_REPO_ROOT/tools/gemm_tn_bench.py > /tmp/pmc_$name.log 2>&1
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python - "$f" "$name" <<'PY' >> $OUT
import csv, sys, collections
f, name = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for row in csv.DictReader(open(f)):
    k = row["Kernel_Name"]
    if "gemm_tn16" not in k: continue
    k = "tn16 grid=" + row.get("Grid_Size", "?")
    acc[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
for k in acc:
    print(name, k, {c: round(v / cnt[(k, c)], 1) for c, v in acc[k].items()}, "dispatches", max(cnt[(k, c)] for c in acc[k]))
PY
  else tail -5 /tmp/pmc_$name.log >> $OUT; fi
}
run sq1 SQ_WAVES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_MFMA
run sq3 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT
# (a FETCH_SIZE / TCC pass of this command did not finish inside 10 minutes on the pool: left out)
cat $OUT
