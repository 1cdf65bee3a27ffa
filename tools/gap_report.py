"""Idle gaps between consecutive kernels of a rocprofv3 --kernel-trace CSV.  Usage: python tools/gap_report.py <kernel_trace.csv> [min_gap_us]"""
import csv, sys, collections

def main():
    path = sys.argv[1]; min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]))
    rows.sort()
    # one steady-state step: between the last two launches of the marker kernel (one per forward)
    marker = sys.argv[3] if len(sys.argv) > 3 else "cls_attention_kernel"
    marks = [r[1] for r in rows if marker in r[2]]
    rows = [r for r in rows if marks[-2] < r[0] <= marks[-1]] if len(marks) >= 2 else rows
    busy = sum(e - s for s, e, _ in rows) / 1e3
    span = (rows[-1][1] - rows[0][0]) / 1e3
    gaps = collections.Counter(); gsum = collections.Counter(); big = []
    prev_end = rows[0][1]; prev_name = rows[0][2]
    for s, e, n in rows[1:]:
        g = (s - prev_end) / 1e3
        if g > 0:
            key = f"{prev_name[:40]} -> {n[:40]}"
            gaps[key] += 1; gsum[key] += g
            if g > min_gap: big.append((g, key))
        if e > prev_end: prev_end, prev_name = e, n
    print(f"span {span:.0f} us, kernel busy {busy:.0f} us, idle {span - busy:.0f} us ({100 * (span - busy) / span:.1f} %), {len(rows)} launches")
    union_idle = sum(gsum.values())
    print(f"no kernel running at all (concurrent streams counted once): {union_idle:.0f} us of {span:.0f} us ({100 * union_idle / span:.1f} %)")
    print("-- largest total idle by transition")
    for k, v in gsum.most_common(25): print(f"{v:9.1f} us  n={gaps[k]:4d}  avg {v / gaps[k]:7.1f}  {k}")
    print("-- single gaps above", min_gap, "us:", len(big), "total", round(sum(g for g, _ in big)), "us")
    for g, k in sorted(big, reverse=True)[:15]: print(f"{g:9.1f} us  {k}")

if __name__ == "__main__":
    main()
