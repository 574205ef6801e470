"""Kernels (and memory copies, when the trace holds them) of the last N events of a rocprofv3 rocpd database in start
order: start offset, duration, gap to the previous event's end.  usage: python scripts/rocpd_timeline.py results.db [N]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, start, end from kernels").fetchall()
try:
    mcols = [r[1] for r in cur.execute("pragma table_info(memory_copies)")]
    size_col = next((c for c in ("size", "bytes") if c in mcols), None)
    mname = "name" if "name" in mcols else None
    sel = f"select {mname or chr(39) + 'copy' + chr(39)}, start, end{', ' + size_col if size_col else ''} from memory_copies"
    for r in cur.execute(sel).fetchall():
        rows.append((f"COPY {r[0]} {r[3] if size_col else ''} B", r[1], r[2]))
except sqlite3.Error as e:
    print("no memory copies:", e)
rows.sort(key=lambda r: r[1])
rows = rows[-n:]
t0, prev_end = rows[0][1], rows[0][1]
for name, s, e in rows:
    print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:7.1f} us  {name[:80]}")
    prev_end = max(prev_end, e)
print(f"span {(prev_end - t0) / 1e3:.1f} us")
