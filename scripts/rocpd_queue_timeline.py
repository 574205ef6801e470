"""One hardware queue of a rocprofv3 rocpd database, kernel by kernel: wait behind the predecessor on the queue, duration, name —
a stretch from the middle of the run.  usage: python scripts/rocpd_queue_timeline.py results.db [kernels to print] [queue rank]"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rank = int(sys.argv[3]) if len(sys.argv) > 3 else 3
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
qcol = next((c for c in ("queue_id", "queue", "stream_id", "stream") if c in cols), None)
rows = cur.execute(f"select {qcol}, {name_col}, start, end from kernels order by {qcol}, start").fetchall()
per_q = collections.defaultdict(list)
for q, name, s, e in rows:
    per_q[q].append((s, e, name))
queues = sorted(per_q, key=lambda q: -len(per_q[q]))
q = queues[min(rank, len(queues) - 1)]
ks = per_q[q]
mid = len(ks) // 2
t0 = ks[mid][0]
print(f"queue {q}: {len(ks)} kernels; {count} of them from the middle (us: start since the first shown, wait behind the predecessor, duration)")
for i in range(mid, min(mid + count, len(ks))):
    s, e, name = ks[i]
    print(f"{(s - t0) / 1e3:10.1f} {(s - ks[i - 1][1]) / 1e3:9.1f} {(e - s) / 1e3:9.1f}  {name.split('(')[0][-44:]}")
