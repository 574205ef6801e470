"""Ad-hoc per-kernel timing probe (not the bench contract): python scripts/probe_timing.py W H depth kind"""
import sys, time
import torch
sys.path.insert(0, ".")
from hydrium_amd import device, synth

w, h, depth, kind = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
img = synth.make_image(kind, w, h, depth, device="cuda")
torch.cuda.synchronize()
n = (-(-w // 2048)) * (-(-h // 2048))
ctx = device.DeviceContext(0, n, 0)
print("best XYB mode proven exact:", ctx.xyb_mode())
for luts in range(ctx.xyb_mode(), 3):
    ctx.set_xyb_mode(luts)
    ctx.encode_image_tensor(img); ctx.sync()
    ctx.profile(True)
    for _ in range(reps):
        ctx.encode_image_tensor(img)
    ctx.sync()
    prof = ctx.profile_read()
    ctx.profile(False)
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.encode_image_tensor(img)
        ctx.sync()
    dt = (time.perf_counter() - t0) / reps
    print(f"{kind} {w}x{h} u{depth} xyb_mode={luts}: e2e {dt*1e3:.3f} ms/frame = {w*h/dt/1e6:.0f} Mpx/s, payload {ctx.payload_size()} B")
    for k, (ms, cnt) in prof.items():
        print(f"   {k:20s} {ms/reps:9.3f} ms/frame  ({cnt//reps} launches, {ms/max(cnt,1)*1e3:8.1f} us each)")
    syms = sum(int(ctx.read_symbol_counts(s).sum()) for s in range(n))
    print(f"   symbols/px {syms/(w*h):.3f}, max group symbols {max(int(ctx.read_symbol_counts(s).max()) for s in range(n))}")
