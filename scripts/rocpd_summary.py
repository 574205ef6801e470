"""Summarise a rocprofv3 rocpd database (kernel-trace --stats) into a small text table.
usage: python scripts/rocpd_summary.py results.db > profiles/rNN_kernel_stats.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                   f"from kernels group by {name_col} order by 3 desc").fetchall()
total = sum(r[2] for r in rows) or 1
print(f"{'kernel':72s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
for n, c, t, a, mn, mx in rows:
    short = n if len(n) <= 72 else n[:69] + "..."
    print(f"{short:72s} {c:6d} {t/1e6:10.3f} {a/1e3:10.1f} {mn/1e3:10.1f} {mx/1e3:10.1f} {100*t/total:6.1f}")
