"""Per-dispatch durations of the EdgeConv kernels from a rocprofv3 --kernel-trace csv: python tools/ws_dispatches.py <kernel_trace.csv>
(the four launches of a step -- tpl / geo graph of the motion pass, tpl / geo of the head -- averaged over the steps)."""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
for key in ("edge_ws_kernel", "edge_rl128_kernel", "split_boundary_rows", "init_boundary_rows"):
    d = [(int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"]) for r in rows if key in r["Kernel_Name"]]
    d.sort()
    if not d:
        continue
    per = {"edge_ws_kernel": 4, "edge_rl128_kernel": 4}.get(key)
    if per:
        n = len(d) // per
        cols = [[d[i * per + j][1] for i in range(n)] for j in range(per)]
        print(key, d[0][2][:40], " ".join(f"{sum(c[1:]) / max(1, len(c) - 1):9.1f}" for c in cols), "us (launch 0..3 of a step, mean without the first step)")
    else:
        print(key, len(d), "launches", f"{sum(x[1] for x in d) / len(d):7.1f} us avg", f"{sum(x[1] for x in d):9.1f} us total")
