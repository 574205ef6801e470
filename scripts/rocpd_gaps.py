"""Per hardware queue of a rocprofv3 rocpd database: for every kernel name, the mean wait between the end of the
previous kernel on the same queue and its own start, and its mean duration — where a frame's stream spends its
latency in the pipelined loop.  usage: python scripts/rocpd_gaps.py results.db [skip_first_n_per_queue]"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
qcol = next((c for c in ("queue_id", "queue", "stream_id", "stream") if c in cols), None)
print("columns:", cols)
rows = cur.execute(f"select {qcol}, {name_col}, start, end from kernels order by {qcol}, start").fetchall()
gap, dur, cnt = collections.Counter(), collections.Counter(), collections.Counter()
per_q = collections.Counter()
prev_q, prev_end = None, None
for q, name, s, e in rows:
    per_q[q] += 1
    if q == prev_q and per_q[q] > skip and ("k_" in name or "rocclr" in name):
        key = name.split("(")[0][-40:]
        gap[key] += max(0, s - prev_end)
        dur[key] += e - s
        cnt[key] += 1
    prev_q, prev_end = q, e
print(f"{'kernel':42s} {'n':>6s} {'wait_us':>9s} {'run_us':>9s}")
tw = tr = 0
for k in sorted(cnt, key=lambda k: -gap[k]):
    print(f"{k:42s} {cnt[k]:6d} {gap[k] / cnt[k] / 1e3:9.1f} {dur[k] / cnt[k] / 1e3:9.1f}")
print("queues:", len(per_q), dict(list(per_q.items())[:24]))
