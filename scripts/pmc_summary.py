"""Summarise rocprofv3 --pmc rocpd database(s): per-kernel mean of each counter.
usage: python scripts/pmc_summary.py results.db [kernel-substring]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
view = "counters_collection" if "counters_collection" in tabs else None
if not view:
    print("no counters_collection view; tables:", [t for t in tabs if "pmc" in t or "counter" in t])
    sys.exit(1)
cols = [r[1] for r in cur.execute(f"pragma table_info({view})")]
kcol = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c and "name" in c][0]
ncol = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
vcol = "value" if "value" in cols else "counter_value"
dcol = "dispatch_id" if "dispatch_id" in cols else None
rows = cur.execute(f"select {kcol}, {ncol}, {('count(distinct ' + dcol + ')') if dcol else 'count(*)'}, sum({vcol}) "
                   f"from {view} where {kcol} like ? group by {kcol}, {ncol} order by 1, 2", (f"%{flt}%",)).fetchall()
last = None
for k, n, c, v in rows:
    if k != last:
        print(f"\n{k[:110]}")
        last = k
    print(f"   {n:34s} dispatches {c:4d}   mean per dispatch {v / max(c, 1):18.1f}")
