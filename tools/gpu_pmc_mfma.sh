#!/bin/bash
# Counter-measured MFMA utilisation of every kernel of the bench command: SQ counters in their
# own rocprofv3 --pmc passes, kernel-trace only (MI355X_MICROARCH.md "rocprofv3 PMC slots": 8 SQ slots per pass).
#   mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES)    busy cycles of the matrix pipes / (4 SIMDs x CU-busy cycles)
# (SQ_VALU_MFMA_BUSY_CYCLES counts cycles, 32 per v_mfma_f32_32x32x16_f16: the same guide's constants table.)
# Output: gpurun_out/mfma_pmc_<tag>.json (copy into profiles/ as mfma_pmc_latest.json; bench.py reads it).
mkdir -p gpurun_out
TAG=${1:-r02}
export TMPDIR=/tmp
cd /tmp
pass() { name=$1; shift
  MORIG_BENCH_NPROC=1 timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmcm_$name -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --prof-steps 0 --secondary 0 --cpu-seconds 0 --batch ${BATCH:-64} > /tmp/pmcm_$name.log 2>&1
  echo "$name pass rc=$? $(tail -c 200 /tmp/pmcm_$name.log | tr '\n' ' ')" | cut -c1-300
}
pass busy SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES
pass inst SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU
pass wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
pass clk GRBM_GUI_ACTIVE
python - "$TAG" <<'PY'
import csv, glob, json, sys, collections, os
tag = sys.argv[1]
out = collections.defaultdict(dict)
for name in ("busy", "inst", "wait", "clk"):
    fs = glob.glob(f"/tmp/pmcm_{name}/**/*counter_collection.csv", recursive=True)
    if not fs:
        continue
    acc, n = collections.defaultdict(collections.Counter), collections.defaultdict(collections.Counter)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "morig" not in k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
    for k in acc:
        for c in acc[k]:
            out[k][c] = acc[k][c] / n[k][c]                    # average per dispatch
            out[k]["dispatches"] = n[k][c]
for k, v in out.items():
    if v.get("SQ_BUSY_CU_CYCLES"):
        v["mfma_util"] = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * v["SQ_BUSY_CU_CYCLES"])
    if v.get("SQ_LDS_IDX_ACTIVE"):
        v["lds_conflict_frac"] = v.get("SQ_LDS_BANK_CONFLICT", 0.0) / v["SQ_LDS_IDX_ACTIVE"]
import hashlib
sha = hashlib.sha256(open(os.environ.get("MORIG_HIP_LIB") or os.path.join(os.environ["GRAFT_REPO_ROOT"], "morig_amd", "lib", "libmorig_hip.so"), "rb").read()).hexdigest()
res = {"lib_sha256": sha, "batch": int(os.environ.get("BATCH", "64")), "note": "rocprofv3 --pmc passes over bench.py --steps 1 --warmup 1 (per-dispatch averages, all XCDs summed by rocprofv3); "
               "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES)", "kernels": out}
json.dump(res, open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", f"mfma_pmc_{tag}.json"), "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0))[:12]:
    print(k[:60], {a: (round(b, 4) if b < 10 else round(b)) for a, b in v.items() if a in ("mfma_util", "lds_conflict_frac", "SQ_BUSY_CU_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F16", "dispatches")})
PY
