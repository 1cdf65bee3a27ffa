#!/bin/bash
# memory-side counters for the hot kernels (separate passes; kernel-trace only)
mkdir -p gpurun_out
TAG=${1:-x}
export TMPDIR=/tmp
cd /tmp
run() {
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$name -- python $GRAFT_REPO_ROOT/tools/microbench.py f16x3 16 > /tmp/pmc_$name.log 2>&1
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python - "$f" "$name" <<'PY' >> $GRAFT_REPO_ROOT/gpurun_out/pmc2_${TAG}.txt
import csv, sys, collections
f, name = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(f)))
# per dispatch id: kernel + counters
disp = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"]
    if "tile_kernel" not in k: continue
    d = disp.setdefault(r["Dispatch_Id"], {"k": k.split("(")[0].replace("void morig::", ""), "grid": r.get("Grid_Size", "")})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
seen = collections.Counter()
for did, d in disp.items():
    key = (d["k"], d["grid"])
    seen[key] += 1
    if seen[key] == 2:      # second dispatch of each (kernel, grid): warm
        print(name, d["k"], "grid", d["grid"], {c: v for c, v in d.items() if c not in ("k", "grid")})
PY
  else tail -5 /tmp/pmc_$name.log >> $GRAFT_REPO_ROOT/gpurun_out/pmc2_${TAG}.txt; fi
}
: > $GRAFT_REPO_ROOT/gpurun_out/pmc2_${TAG}.txt
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
cat $GRAFT_REPO_ROOT/gpurun_out/pmc2_${TAG}.txt
