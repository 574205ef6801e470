import time, sys, ctypes, hashlib
sys.path.insert(0, '.')
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
