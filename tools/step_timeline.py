"""One steady-state step of a rocprofv3 --kernel-trace CSV as a timeline: start offset, duration, queue of every kernel.
usage: python tools/step_timeline.py <kernel_trace.csv> <marker kernel substring (one launch per step)>"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]))
rows.sort()
marker = sys.argv[2]
marks = [r[0] for r in rows if marker in r[3]]
lo, hi = marks[-2], marks[-1]
step = [r for r in rows if lo <= r[0] < hi]
t0 = step[0][0]
print(f"step span {(max(r[1] for r in step) - t0) / 1e3:.0f} us, {len(step)} launches")
for s, e, q, n in step:
    n = n.replace("void morig::", "").replace("morig::", "")
    print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:8.1f} us  q{q:>3}  {n[:80]}")
