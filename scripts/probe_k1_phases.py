#!/usr/bin/env python3
"""Where do the transform kernel's cycles go?  Builds libhydrium with -DHYDK_PHASE_TIMERS (s_memtime
stamps around the phases of k_transform_tokenize, summed over every wave of a launch) and prints
each phase's share for one 8192x8192 RGB16 photo frame.

    python scripts/probe_k1_phases.py --build     # here (hipcc cross-compiles)
    python scripts/probe_k1_phases.py --run       # on the GPU box
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "scripts", "probe_build")
LIB = os.path.join(OUT, "libhydrium_phases.so")
PHASES = ["A1 pixels -> XYB (LUT evaluation, LMS mix)", "A2 row DCT + LDS store", "barrier 1", "B column DCT, quantise, bitmaps, LF ints",
          "barrier 2", "C1 prefix sum (+ barrier 3)", "C2 token emission", "prologue / epilogue"]


def build(define="-DHYDK_PHASE_TIMERS", lib=None):
    from hydrium_amd import build as hb

    lib = lib or LIB
    os.makedirs(OUT, exist_ok=True)
    objs = []
    for src in sorted(os.listdir(os.path.join(hb.CSRC, "hip"))):
        if src.endswith(".hip"):
            o = os.path.join(OUT, src + ".o")
            hb._run([hb.HIPCC] + hb.HIP_FLAGS + [define, "-c", os.path.join(hb.CSRC, "hip", src), "-o", o])
            objs.append(o)
    for src in sorted(os.listdir(os.path.join(hb.CSRC, "host"))):
        if src.endswith(".c"):
            o = os.path.join(OUT, src + ".o")
            hb._run([hb.CC] + hb.C_FLAGS + ["-c", os.path.join(hb.CSRC, "host", src), "-o", o])
            objs.append(o)
    hb._run([hb.HIPCC, f"--offload-arch={hb.ARCH}", "-shared", "-fPIC", "-o", lib] + objs + ["-lpthread"])
    print(lib)


def run():
    os.environ["HYDAMD_LIB"] = LIB
    import torch

    from hydrium_amd import device, synth

    img = synth.make_image("photo", 8192, 8192, 16, device=torch.device("cuda", 0))
    ctx = device.DeviceContext(0, 16, 0)
    ctx.set_lf_coder(0)
    ticks = (C.c_ulonglong * 8)()
    for rep in range(3):
        ctx.encode_image_tensor(img)
        ctx.sync()
        ctx.d.hydamd_debug_phase_ticks(ticks, 1)
    tot = float(sum(ticks))
    print("k_transform_tokenize, 8192x8192 RGB16 photo, s_memtime ticks summed over all waves (third run):")
    for name, t in zip(PHASES, ticks):
        print(f"  {name:48s} {t:16d}  {100.0 * t / tot:5.1f} %")
    ctx.close()


if __name__ == "__main__":
    if "--build" in sys.argv:
        build()
    if "--build-variant" in sys.argv:  # --build-variant -DNAME=VALUE out.so : any other compile-time variant of the library
        i = sys.argv.index("--build-variant")
        build(sys.argv[i + 1], os.path.join(OUT, sys.argv[i + 2]))
    if "--run" in sys.argv:
        run()
