"""Eight 8192x8192 RGB16 frames through the drop-in API, one at a time: per-frame wall time through the ctypes caller
(the command rocprofv3 wraps for the API path's kernel trace)."""
import time, sys, ctypes, hashlib, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hydrium_amd import api, synth
img = synth.make_image("photo", 8192, 8192, 16, device="cuda").cpu().numpy().view(np.uint16)
img = np.ascontiguousarray(img)
lib = api.Library()
big = (ctypes.c_uint8 * (32 << 20))()
for i in range(8):
    t = time.perf_counter()
    d = api.encode_image(lib, img, out_buf=big)
    print(i, round((time.perf_counter() - t) * 1e3, 2), len(d), hashlib.md5(d).hexdigest()[:8], flush=True)
