"""Standalone duration of the LF-group coder kernels: transform stage only, so nothing competes for CUs.
usage: python scripts/lf_timing.py [W H depth kind]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hydrium_amd import device as dev, synth
W, H, D = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (8192, 8192, 16)
kind = sys.argv[4] if len(sys.argv) > 4 else "photo"
img = synth.make_image(kind, W, H, D, device="cuda")
lfx, lfy = (W + 2047) // 2048, (H + 2047) // 2048
n = lfx * lfy
with dev.DeviceContext(0, n) as c:
    for on in (1, 0, 1):
        c.set_lf_coder(bool(on))
        c.encode_image_tensor(img); c.sync()
        c.profile(True)
        t0 = time.perf_counter()
        for _ in range(5):
            c.begin_frame(n)
            isz = img.element_size()
            for s in range(n):
                x0, y0 = (s % lfx) * 2048, (s // lfx) * 2048
                w, h = min(2048, W - x0), min(2048, H - y0)
                base = img.data_ptr() + (y0 * W + x0) * 3 * isz
                c.encode_lf_group(s, [base, base + isz, base + 2 * isz], 3 * W, 3, 0 if D == 8 else 1, w, h, s)
            c.run_transform(n)
            c.sync_streams() if hasattr(c, "sync_streams") else torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print("lf_coder", on, "transform-only frame %.3f ms" % (dt * 1e3), {k: round(ms / max(cnt, 1), 4) for k, (ms, cnt) in c.profile_read().items() if cnt})
        c.profile(False)
