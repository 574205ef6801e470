"""How many kernels of each class run at the same time in the steady state of a pipelined run (rocprofv3 rocpd database):
time-averaged number of concurrently running launches per kernel name, taken over the window between the 40th and
the 90th percentile of the transform kernel's launches; each class's mean duration; and one queue's timeline.
usage: python scripts/rocpd_concurrency.py results.db [queue index to print]"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = cur.execute("select name, start, end, queue_id from kernels order by start").fetchall()


def short(name):
    return name.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")[:44] or "(unnamed)"


k1 = [r for r in rows if "k_transform_tokenize" in r[0]]
lo, hi = k1[int(len(k1) * 0.4)][1], k1[int(len(k1) * 0.9)][1]
busy, n, dur = collections.Counter(), collections.Counter(), collections.Counter()
for name, s, e, q in rows:
    key = short(name)
    ov = min(e, hi) - max(s, lo)
    if ov > 0:
        busy[key] += ov
    if lo <= s < hi:
        n[key] += 1
        dur[key] += e - s
span = hi - lo
print(f"window {span / 1e6:.2f} ms")
print(f"{'kernel':46s} {'launches':>8s} {'mean_us':>9s} {'concurrent':>10s}")
for k in sorted(busy, key=lambda k: -busy[k]):
    print(f"{k:46s} {n[k]:8d} {dur[k] / max(n[k], 1) / 1e3:9.1f} {busy[k] / span:10.2f}")
frames = sum(v for k, v in n.items() if "transform_tokenize" in k)
print(f"frames in window: {frames} -> {span / frames / 1e6:.4f} ms per frame")
qs = sorted({r[3] for r in k1})
q = qs[int(sys.argv[2]) if len(sys.argv) > 2 else len(qs) // 2]
print(f"\ntimeline of queue {q} inside the window (start offset, duration, gap to the previous kernel's end on this queue):")
prev = None
for name, s, e, qq in rows:
    if qq != q or s < lo or s > lo + (hi - lo) * 0.45:
        continue
    print(f"  {(s - lo) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f}  gap {((s - prev) / 1e3 if prev else 0):8.1f}  {short(name)}")
    prev = e
