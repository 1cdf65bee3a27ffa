#!/bin/bash
# HBM traffic of every kernel of the bench command from the L2 memory-side counters (MI355X_MICROARCH.md, HBM):
# separate --pmc passes (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2), kernel-trace only. Output: per kernel the
# average per-dispatch FETCH_SIZE / WRITE_SIZE in KiB -> gpurun_out/traffic_<tag>.json (copy into profiles/).
mkdir -p gpurun_out
TAG=${1:-r01}
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  MORIG_BENCH_NPROC=1 timeout 420 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmcb_$C -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --cpu-seconds 0 --secondary 0 --prof-steps 0 --batch ${BATCH:-64} > /tmp/pmcb_$C.log 2>&1
  echo "$C pass rc=$? $(tail -c 300 /tmp/pmcb_$C.log | tr '\n' ' ')" | cut -c1-400
done
python - "$TAG" <<'PY'
import csv, glob, json, sys, collections, os
tag = sys.argv[1]
out = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(f"/tmp/pmcb_{c}/**/*counter_collection.csv", recursive=True)
    if not fs:
        continue
    acc, n = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] != c:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k] += float(r["Counter_Value"]); n[k] += 1
    for k in acc:
        out[k][c + "_KiB_per_dispatch"] = acc[k] / n[k]
        out[k]["dispatches"] = n[k]
import hashlib
sha = hashlib.sha256(open(os.environ.get("MORIG_HIP_LIB") or os.path.join(os.environ["GRAFT_REPO_ROOT"], "morig_amd", "lib", "libmorig_hip.so"), "rb").read()).hexdigest()
res = {"lib_sha256": sha, "batch": int(os.environ.get("BATCH", "64")), "note": "rocprofv3 --pmc, bench.py --steps 1 --warmup 1; FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 "
               "(MI355X_MICROARCH.md HBM): hbm_bytes = (2*FETCH + WRITE) * 1024", "kernels": out}
json.dump(res, open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", f"traffic_{tag}.json"), "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("FETCH_SIZE_KiB_per_dispatch", 0))[:12]:
    print(k[:70], {a: round(b, 1) for a, b in v.items()})
PY
