"""BASELINE config C5 through the drop-in API: a batch of independent 3840x2160 RGB8 frames, one
encoder per frame, T host threads (ctypes releases the GIL inside the library).
usage: python scripts/api_batch.py [frames] [threads...]"""
import os, sys, time, threading, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hydrium_amd import api, synth

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 64
thread_counts = [int(a) for a in sys.argv[2:]] or [1, 2, 4, 8]
w, h = 3840, 2160
imgs = [synth.make_image("photo", w, h, 8, seed=100 + i) for i in range(8)]
lib = api.Library()
want = [hashlib.md5(api.encode_image(lib, im)).hexdigest() for im in imgs]  # also warms the library
for T in thread_counts:
    out = [None] * frames
    def work(t):
        for f in range(t, frames, T):
            out[f] = hashlib.md5(api.encode_image(lib, imgs[f % 8])).hexdigest()
    ts = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    dt = time.perf_counter() - t0
    ok = all(out[f] == want[f % 8] for f in range(frames))
    print(f"{T} threads: {frames} frames in {dt*1e3:.0f} ms = {frames/dt:.0f} frames/s = {frames*w*h/dt/1e6:.0f} Mpixel/s, "
          f"{dt/frames*1e3:.2f} ms per frame, bytes identical to the single-thread run: {ok}", flush=True)
