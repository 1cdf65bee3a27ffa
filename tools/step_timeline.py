"""One steady-state step of a rocprofv3 --kernel-trace CSV as a timeline: start offset, duration, queue of every kernel.
usage: python tools/step_timeline.py <kernel_trace.csv> <marker kernel substring (one launch per step)>"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        grid = "x".join(str(r.get(k, "?")) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")) if "Grid_Size_X" in r else r.get("Grid_Size", "?")
        wg = r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"], grid, wg))
rows.sort()
marker = sys.argv[2]
marks = [r[0] for r in rows if marker in r[3]]
lo, hi = marks[-2], marks[-1]
step = [r for r in rows if lo <= r[0] < hi]
t0 = step[0][0]
print(f"step span {(max(r[1] for r in step) - t0) / 1e3:.0f} us, {len(step)} launches")
gap = 0
prev_end = t0
for s, e, q, n, grid, wg in step:
    gap += max(0, s - prev_end)
    prev_end = max(prev_end, e)
    n = n.replace("void morig::", "").replace("morig::", "")
    print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:8.1f} us  q{q:>3}  grid {grid:>14} wg {wg:>4}  {n[:80]}")
print(f"idle between launches: {gap / 1e3:.0f} us")
