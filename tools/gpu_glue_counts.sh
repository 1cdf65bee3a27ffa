#!/bin/bash
# How many torch glue launches (blit copies, fills) does ONE steady-state step issue? rocprofv3 kernel stats of the bench command at two
# step counts (no secondary workloads, no CPU leg): whatever does not scale with the step count is set-up (parameter upload: one blit
# per state_dict entry, weight packing), not the step.   usage: tools/gpu_glue_counts.sh <tag>
cd /tmp && export TMPDIR=/tmp
TAG=${1:-glue}
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
for K in 5 25; do
  rm -rf /tmp/glue_$K
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/glue_$K -- python $GRAFT_REPO_ROOT/bench.py --steps $K --warmup 2 --cpu-seconds 0 --secondary 0 --prof-steps 0 > /tmp/glue_$K.log 2>&1
  f=$(find /tmp/glue_$K -name "*kernel_stats.csv" | head -1)
  cp "$f" $OUT/glue_stats_${TAG}_$K.csv
done
python - <<PY
import csv
def load(k):
    return {r['Name']: int(r['Calls']) for r in csv.DictReader(open('$OUT/glue_stats_${TAG}_%d.csv' % k))}
a, b = load(5), load(25)
names = sorted(set(a) | set(b), key=lambda n: -(b.get(n, 0)))
lines = ["launches of the bench process at 5 and at 25 timed steps (2 warm-up each): per-step count = (b - a) / 20, set-up = a - 7 * per-step"]
tot_step = tot_setup = 0
for n in names:
    ca, cb = a.get(n, 0), b.get(n, 0)
    per = (cb - ca) / 20.0
    setup = ca - 7 * per
    if 'morig::' in n:
        continue
    tot_step += per; tot_setup += setup
    if cb >= 20:
        lines.append(f"{per:8.2f} per step  {setup:8.0f} set-up   {n[:110]}")
lines.append(f"torch / runtime launches (everything outside morig::): {tot_step:.1f} per step, {tot_setup:.0f} at set-up")
open('$OUT/glue_counts_$TAG.txt', 'w').write("\n".join(lines) + "\n")
print("\n".join(lines[-12:]))
PY
