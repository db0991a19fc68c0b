"""Per-kernel summary of a rocprofv3 run that wrote a rocpd database (the default output of rocprofv3 7.x): calls, average /
minimum duration, share of the GPU time, and -- for the GEMMs -- the averages by grid size.
    python tools/rocpd_stats.py <results.db> [name filter]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = db.execute("select name, count(*), avg(end-start), min(end-start), sum(end-start) from kernels group by name order by 5 desc").fetchall()
tot = sum(r[4] for r in rows)
print(f"{'kernel':64s} {'calls':>6s} {'avg us':>9s} {'min us':>9s} {'share':>6s}")
for name, n, avg, mn, s in rows:
    if flt in name and s / tot > 0.001:
        print(f"{name.split('(')[0].replace('void ', '')[:64]:64s} {n:6d} {avg / 1e3:9.1f} {mn / 1e3:9.1f} {100 * s / tot:5.1f}%")
print("by grid size:")
for name, gx, n, avg in db.execute("select name, grid_x, count(*), avg(end-start) from kernels where name like '%gemm%' group by name, grid_x"):
    print(f"  {name.split('(')[0].replace('void ', '')[:40]:40s} grid_x {gx:9d} {n:5d} calls {avg / 1e3:9.1f} us")
