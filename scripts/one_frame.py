"""A few 8192x8192 RGB16 photo frames, one at a time, through the device-level API: the command
rocprofv3 wraps for per-kernel traces and PMC passes (scripts/collect_pmc.sh).
usage: python scripts/one_frame.py [frames] [rans form] [lf coder mode] [times]
(a fourth argument prints every stage's time by the library's own event timers)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from hydrium_amd import device, synth

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 3
form = int(sys.argv[2]) if len(sys.argv) > 2 else 5
lf = int(sys.argv[3]) if len(sys.argv) > 3 else 2
img = synth.make_image("photo", 8192, 8192, 16, device=torch.device("cuda", 0))
with device.DeviceContext(0, 16, 0) as ctx:
    ctx.set_rans_waves(form)
    ctx.set_lf_coder(lf)
    for _ in range(frames):
        ctx.encode_image_tensor(img)
        ctx.sync()
    print(ctx.payload_size(), "section bytes")
    if len(sys.argv) > 4:
        ctx.profile(True)
        for _ in range(8):
            ctx.encode_image_tensor(img)
            ctx.sync()
        for k, (ms, n) in ctx.profile_read().items():
            if n:
                print(f"   {k:24s} {ms / n:.4f} ms")
