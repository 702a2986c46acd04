r"""Turns a rocprofv3 ``*_results.db`` (rocpd sqlite, ROCm 7.2 default output) into the per-kernel
``--stats`` table committed under profiles/.   python profiles/summarize.py <db> [top]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows = list(db.execute(
    "select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
    "from kernels group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print(f"{'kernel':78s} {'calls':>7s} {'total_us':>13s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>10s} {'pct':>6s}")
for r in rows[:top]:
    print(f"{r[0][:78]:78s} {r[1]:7d} {r[2]:13.1f} {r[3]:10.2f} {r[4]:9.2f} {r[5]:10.2f} {100 * r[2] / tot:6.2f}")
print(f"{'TOTAL':78s} {sum(r[1] for r in rows):7d} {tot:13.1f}")
