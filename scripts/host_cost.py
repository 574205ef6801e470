import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
from hydrium_amd import device, synth
img = synth.make_image("photo", 8192, 8192, 16, device="cuda")
ctxs = [device.DeviceContext(0, 16) for _ in range(12)]
for c in ctxs:
    c.set_rans_waves(5); c.set_lf_coder(2); c.encode_image_tensor(img)
for c in ctxs: c.sync()
# host cost of enqueuing one frame when the GPU is idle
ts = []
for i in range(24):
    c = ctxs[i % 12]
    c.sync()
    t = time.perf_counter(); c.encode_image_tensor(img); ts.append(time.perf_counter() - t)
for c in ctxs: c.sync()
ts.sort()
print("enqueue one frame: median %.3f ms, min %.3f, max %.3f" % (ts[12] * 1e3, ts[0] * 1e3, ts[-1] * 1e3))
