import os, sys, time
os.environ["HYDAMD_TRACE"] = "1"
sys.path.insert(0, ".")
import numpy as np
from hydrium_amd import api, synth
w, h, d = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (8192, 8192, 16)
img = synth.make_image("photo", w, h, d, device="cuda").cpu().numpy()
img = np.ascontiguousarray(img.view(np.uint16) if d == 16 else img)
lib = api.Library()
for rep in range(2):
    t = time.perf_counter(); out = api.encode_image(lib, img); dt = time.perf_counter() - t
    print(f"rep {rep}: {dt*1e3:.1f} ms, {w*h/dt/1e6:.0f} Mpx/s, {len(out)} bytes", file=sys.stderr)
