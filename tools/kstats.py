"""Compact view of a rocprofv3 --kernel-trace --stats run: the library's own kernels (rr_* / ce_*) from *_kernel_stats.csv.
    python tools/kstats.py <dir> [min_calls]"""
import csv, glob, sys
d = sys.argv[1]
min_calls = int(sys.argv[2]) if len(sys.argv) > 2 else 1
f = sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True))[0]
rows = []
for r in csv.DictReader(open(f)):
    name = r["Name"].replace("void ", "")
    if not (name.startswith("rr_") or name.startswith("ce_")) or int(r["Calls"]) < min_calls:
        continue
    short = name.split("(")[0]
    rows.append((float(r["TotalDurationNs"]), short, int(r["Calls"]), float(r["AverageNs"]), float(r["MinNs"]), float(r["MaxNs"])))
print(f"{'kernel':60s} {'calls':>6s} {'avg us':>9s} {'min us':>9s} {'max us':>9s}")
for tot, n, c, a, mn, mx in sorted(rows, reverse=True):
    print(f"{n[:60]:60s} {c:6d} {a / 1e3:9.1f} {mn / 1e3:9.1f} {mx / 1e3:9.1f}")
