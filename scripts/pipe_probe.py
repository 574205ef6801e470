#!/usr/bin/env python3
"""What bounds the pipelined loop (bench.py --mode frame)?  The same loop, stripped to its core, with knobs:

    python scripts/pipe_probe.py [--streams 16] [--frames 192] [--profile 0|1] [--lf 0|2] [--rans 5] [--reps 3]

Prints, per repetition: frames/ms by HIP events (last priming frame -> last timed frame), the host's time per frame
spent enqueuing (time inside encode_image_tensor), and how far ahead of the GPU the host ran.  HYDAMD_LIB selects a
library variant (scripts/k1_variants.py)."""
import argparse
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "28")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# the measurement switches (HYDAMD_DEBUG_SKIP, the stand-in kernels) exist in the HYD_TEST_HOOKS flavour of the library only
os.environ.setdefault("HYDAMD_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hydrium_amd", "lib", "libhydrium_probe.so"))
import torch  # noqa: E402

from hydrium_amd import device, placement, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, default=16)
ap.add_argument("--frames", type=int, default=192)
ap.add_argument("--profile", type=int, default=0)
ap.add_argument("--lf", type=int, default=2)
ap.add_argument("--rans", type=int, default=5)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--size", type=int, default=8192)
ap.add_argument("--depth", type=int, default=16, choices=(8, 16))
ap.add_argument("--height", type=int, default=0, help="frame height if not square: a frame of k x 8192 rows stands for k frames coded as one launch group")
ap.add_argument("--no-bind", action="store_true")
ap.add_argument("--only-transform", action="store_true", help="enqueue the transform stage only (no entropy stage, no LF coder)")
ap.add_argument("--batch", type=int, default=1, help="frames per launch group (hydamd_encode_image_batch)")
ap.add_argument("--cohort", type=int, default=0, help="> 0: drain every context after this many launch groups (does a restart bring the fast first phase back?)")
ap.add_argument("--clock", type=int, default=0, help="> 0: sample the shader clock (hydamd_debug_shader_clock_mhz) every this many launch groups")
ap.add_argument("--events", default="torch", choices=("torch", "device", "none"),
                help="the probe's own event per launch group: torch (timing events, system-scope release: the default until now), "
                     "device (HIP events created with hipEventReleaseToDevice), none (the rate is the run's wall clock)")
ap.add_argument("--chain-clock", action="store_true", help="with a library built under -DHYDK_CHAIN_PROBE=32 (and HYDAMD_DEBUG_SKIP=4): when did the chain wavefronts of context 0's last launch group start and end?")
ap.add_argument("--lanes", type=int, default=0, help="> 0: this many HIP streams, contexts dealt to them in turn (several contexts per stream)")
ap.add_argument("--high", type=int, default=-1, help=">= 0: every context on a torch stream of its own, the first this many of them created with HIGH "
                                                    "priority (does a staggered mix of stages beat sixteen streams progressing alike?)")
a = ap.parse_args()
if not a.no_bind:
    placement.bind_near_gpu(0)
H = a.height or a.size
img = synth.make_image("photo", a.size, H, a.depth, device=torch.device("cuda", 0))
lfg = (-(-a.size // 2048)) * (-(-H // 2048))
ctxs = [device.DeviceContext(0, lfg * a.batch, 0) for _ in range(a.streams)]
if a.batch > 1:
    for c in ctxs:
        c.encode_image_tensor = lambda im, c=c: c.encode_image_batch([im] * a.batch)
if a.lanes:
    lanes = [torch.cuda.Stream() for _ in range(a.lanes)]
    for i, c in enumerate(ctxs):
        c.set_stream(lanes[i % a.lanes].cuda_stream)
if a.high >= 0:
    prio = [torch.cuda.Stream(priority=-1 if i < a.high else 0) for i in range(len(ctxs))]
    for c, st in zip(ctxs, prio):
        c.set_stream(st.cuda_stream)
ext = [torch.cuda.ExternalStream(c.get_stream()) for c in ctxs]
for c in ctxs:
    c.set_rans_waves(a.rans)
    c.set_lf_coder(a.lf)
    c.encode_image_tensor(img)
for c in ctxs:
    c.sync()
    c.profile(bool(a.profile))
S = a.streams
import ctypes as _C
try:
    _hip = _C.CDLL("libamdhip64.so.7")  # by soname: the runtime torch already loaded
except OSError:
    _hip = _C.CDLL("libamdhip64.so")


class DevEvent:
    """a HIP event with timing whose record releases at DEVICE scope (hipEventReleaseToDevice = 0x40000000)"""

    def __init__(self, stream):
        self.h = _C.c_void_p()
        assert _hip.hipEventCreateWithFlags(_C.byref(self.h), _C.c_uint(0x40000000)) == 0
        assert _hip.hipEventRecord(self.h, _C.c_void_p(stream)) == 0

    def elapsed_time(self, other):
        ms = _C.c_float()
        assert _hip.hipEventElapsedTime(_C.byref(ms), self.h, other.h) == 0
        return ms.value


for rep in range(a.reps):
    torch.cuda.synchronize()
    host = 0.0
    evs = []
    t0 = time.perf_counter()
    n = 4 * S + a.frames + S
    first = 0.0
    clocks = []
    for i in range(n):
        th = time.perf_counter()
        if a.only_transform:
            c = ctxs[i % S]
            c.per_lf_group_calls = True
            c.finish_frame = lambda n, c=c: c.run_transform(n)
        ctxs[i % S].encode_image_tensor(img)
        host += time.perf_counter() - th
        if i == 3 * S - 1:
            first = host / (3 * S)  # no context has more than three frames queued yet: nothing can have blocked
        if a.events == "torch":
            e = torch.cuda.Event(enable_timing=True)
            e.record(ext[i % S])
            evs.append(e)
        elif a.events == "device":
            evs.append(DevEvent(ctxs[i % S].get_stream()))
        if a.cohort and i % a.cohort == a.cohort - 1:
            for c in ctxs:
                c.sync()
        if a.clock and i % a.clock == a.clock - 1:
            clocks.append(round(ctxs[0].shader_clock_mhz()))
    t_issue = time.perf_counter() - t0
    for c in ctxs:
        c.sync()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    if a.events == "none":
        print(f"batch {a.batch} streams {S} rans {a.rans}, no events: wall clock {a.size * H * a.batch * n / wall / 1e9:.1f} Gpixel/s over the whole run "
              f"({n} launch groups, fill and drain included), host enqueue {host / n * 1e3:.3f} ms per group (unblocked: {first * 1e3:.3f})", flush=True)
        continue
    w = min(S, a.frames)
    base = evs[0]
    t_start = sum(base.elapsed_time(e) for e in evs[4 * S - w:4 * S]) / w
    t_end = sum(base.elapsed_time(e) for e in evs[4 * S + a.frames - w:4 * S + a.frames]) / w
    ms = (t_end - t_start) / a.frames / a.batch
    # the rate over time: completion times of successive chunks of 2 S launch groups
    done = [base.elapsed_time(e) for e in evs]
    chunk = 2 * S
    marks = [sum(done[k - w:k]) / w for k in range(4 * S, len(done) + 1, chunk)]
    trend = " ".join(f"{a.size * H * a.batch * chunk / (marks[j + 1] - marks[j]) / 1e6:.0f}" for j in range(len(marks) - 1))
    if clocks:
        print("   shader clock, MHz, every", a.clock, "launch groups:", " ".join(str(c) for c in clocks), flush=True)
    print(f"   Gpixel/s in successive chunks of {chunk} launch groups: {trend}", flush=True)
    vals = [a.size * H * a.batch * chunk / (marks[j + 1] - marks[j]) / 1e6 for j in range(len(marks) - 1)]
    if len(vals) >= 8:
        mid = sorted(vals[3:-2])
        print(f"   SUSTAINED (median chunk, first three and last two left out): {mid[len(mid) // 2]:.1f} Gpixel/s", flush=True)
    print(f"batch {a.batch} lanes {a.lanes} streams {S} profile {a.profile} lf {a.lf} rans {a.rans}: {ms:.4f} ms/frame = {a.size * H / ms / 1e6:.1f} Gpixel/s; "
          f"host enqueue {host / n * 1e3:.3f} ms/frame (first 3 per stream, unblocked: {first * 1e3:.3f}), issue loop {t_issue / n * 1e3:.3f} ms/frame, wall {wall / n * 1e3:.3f} ms/frame",
          flush=True)
    if a.chain_clock:  # 100 MHz ticks, wrapped to 32 bits: differences only
        import numpy as np
        t = np.array([[int(ctxs[0].read_sections(sl)[0][k]) for k in (0, 1)] for sl in range(lfg * a.batch)], dtype=np.int64)
        s0 = (t[:, 0] - t[:, 0].min()) / 100.0  # us after the first wavefront's start
        run = ((t[:, 1] - t[:, 0]) % (1 << 32)) / 100.0
        print(f"   chain wavefronts of context 0's last launch group ({len(t)}): started over {s0.max():.0f} us (median {np.median(s0):.0f}), "
              f"each ran {run.min():.0f} .. {run.max():.0f} us (median {np.median(run):.0f}); first start -> last end {((t[:, 1].max() - t[:, 0].min()) % (1 << 32)) / 100.0:.0f} us", flush=True)
    if a.profile:  # every stage's duration as it runs INSIDE the loop, sharing the chip (event timers, context 0)
        print("   stage times in the loop: " + ", ".join(f"{k} {ms / n:.3f} ms" for k, (ms, n) in ctxs[0].profile_read().items() if n), flush=True)
