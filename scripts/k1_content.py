"""K1 alone on different content (does a LUT gather, whose cost depends on how the indices scatter, pay everywhere?)
usage: HYDAMD_LIB=<variant> python scripts/k1_content.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hydrium_amd import device, synth

for kind, depth in (("photo", 16), ("noise", 16), ("smooth", 16), ("photo", 8), ("noise", 8)):
    img = synth.make_image(kind, 8192, 8192, depth, device=torch.device("cuda", 0))
    with device.DeviceContext(0, 16, 0) as ctx:
        ctx.set_rans_waves(5)
        ctx.set_lf_coder(0)
        ctx.encode_image_tensor(img); ctx.sync()
        ctx.profile(True)
        for _ in range(6):
            ctx.encode_image_tensor(img); ctx.sync()
        ms, n = ctx.profile_read()["transform_tokenize"]
        print(f"{kind:7s} u{depth:<2d} K1 {ms / n:.4f} ms", flush=True)
