#!/bin/bash
# kernel-level view of the training step: rocprofv3 kernel stats of (2 warm-up + 3) steps -> gpurun_out/train_prof_$TAG.csv
TAG=${1:-r02}
mkdir -p gpurun_out
export TMPDIR=/tmp
D=$(pwd)
timeout 200 python tools/train_prof.py 8 3
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/tp -o tp --output-format csv -- python $D/tools/train_prof.py 8 3 2>&1 | tail -3
f=$(find /tmp/tp -name '*kernel_stats.csv' | head -1)
cp "$f" $D/gpurun_out/train_prof_$TAG.csv
head -25 $D/gpurun_out/train_prof_$TAG.csv | cut -c1-160
python - <<PY
import csv
rows=list(csv.DictReader(open("$D/gpurun_out/train_prof_$TAG.csv")))
print("kernel time per step: %.1f ms over %d launches/step" % (sum(float(r["TotalDurationNs"]) for r in rows)/5e6, sum(int(r["Calls"]) for r in rows)/5))
PY
