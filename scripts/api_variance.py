"""Where does the hyd_send_tile path spend its time, call by call (hunting the occasional slow frame)?
usage: python scripts/api_variance.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from hydrium_amd import api, synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
w = h = 8192
img = synth.make_image("photo", w, h, 16, device="cuda").cpu().numpy()
img = np.ascontiguousarray(img.view(np.uint16))
lib = api.Library()
for rep in range(reps):
    t = {"new": 0.0, "meta": 0.0, "send": 0.0, "flush": 0.0, "copy": 0.0, "destroy": 0.0}
    sends = []
    t0 = time.perf_counter()
    a = time.perf_counter(); enc = api.Encoder(lib); t["new"] += time.perf_counter() - a
    a = time.perf_counter(); enc.check(enc.set_metadata(w, h)); buf = (C.c_uint8 * (1 << 20))(); enc.check(enc.provide_output(buf)); t["meta"] += time.perf_counter() - a
    out = bytearray()
    for ty in range(4):
        for tx in range(4):
            a = time.perf_counter(); enc.check(enc.send_tile(img, tx, ty, 2048, 2048)); d = time.perf_counter() - a
            t["send"] += d; sends.append(round(d * 1e3, 2))
            while True:
                a = time.perf_counter(); ret = enc.check(enc.flush()); code, n = enc.release_output(); t["flush"] += time.perf_counter() - a
                a = time.perf_counter(); out += C.string_at(buf, n); enc.check(enc.provide_output(buf)); t["copy"] += time.perf_counter() - a
                if ret != api.HYD_NEED_MORE_OUTPUT:
                    break
    a = time.perf_counter(); enc.close(); t["destroy"] += time.perf_counter() - a
    total = time.perf_counter() - t0
    print(f"rep {rep}: {total*1e3:7.1f} ms  " + "  ".join(f"{k} {v*1e3:.1f}" for k, v in t.items()) + f"  sends {sends}", flush=True)
