"""In-tree build of libhydrium.so.0 (host C + HIP kernels for gfx950).

Everything is compiled with -ffp-contract=off: the reference's canonical bytes are the
non-contracted ones.  Objects go to hydrium_amd/build/, the library to hydrium_amd/lib/ (both
git-ignored; the built library travels to the GPU box with the gpurun snapshot).
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ_DIR = os.path.join(HERE, "build")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libhydrium.so.0")
HOSTTEST_PATH = os.path.join(LIB_DIR, "libhydrium_hosttest.so")

ARCH = os.environ.get("HYDAMD_ARCH", "gfx950")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CC = os.environ.get("CC", "gcc")

HIP_FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden",
             "-fno-slp-vectorize",  # packed-f32 chains need a wait state per dependent op on gfx950: slower than scalar
             "-Wall", "-Wno-unused-function", f"-I{os.path.join(ROOT, 'include')}"]
C_FLAGS = ["-std=c99", "-O2", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden", "-Wall", "-Wextra",
           "-Wno-unused-parameter", "-DHYDRIUM_INTERNAL_BUILD", f"-I{os.path.join(ROOT, 'include')}",
           f"-I{os.path.join(CSRC, 'host')}"]


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + r.stderr)
        raise RuntimeError(f"build step failed: {cmd[0]} ... {cmd[-1]}")
    return r


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ_DIR, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    headers = glob.glob(os.path.join(CSRC, "**", "*.h"), recursive=True) + \
        glob.glob(os.path.join(ROOT, "include", "**", "*.h"), recursive=True)
    objs, test_objs = [], []
    for src in sorted(glob.glob(os.path.join(CSRC, "hip", "*.hip"))):
        obj = os.path.join(OBJ_DIR, os.path.basename(src) + ".o")
        if force or _newer(obj, [src] + headers):
            if verbose:
                print("hipcc", os.path.basename(src))
            _run([HIPCC] + HIP_FLAGS + ["-c", src, "-o", obj])
        objs.append(obj)
    for src in sorted(glob.glob(os.path.join(CSRC, "host", "*.c"))):
        obj = os.path.join(OBJ_DIR, os.path.basename(src) + ".o")
        if force or _newer(obj, [src] + headers):
            if verbose:
                print("cc", os.path.basename(src))
            _run([CC] + C_FLAGS + ["-c", src, "-o", obj])
        objs.append(obj)
        # the same host sources with the test hooks visible, for the CPU-only glue tests
        tobj = os.path.join(OBJ_DIR, os.path.basename(src) + ".test.o")
        if force or _newer(tobj, [src] + headers):
            _run([CC] + C_FLAGS + ["-DHYD_TEST_HOOKS", "-c", src, "-o", tobj])
        test_objs.append(tobj)
    if force or _newer(LIB_PATH, objs):
        _run([HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-Wl,-soname,libhydrium.so.0", "-o", LIB_PATH] + objs + ["-lpthread"])
    if test_objs and (force or _newer(HOSTTEST_PATH, test_objs)):
        hip_objs = [o for o in objs if o.endswith(".hip.o")]
        _run([HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", HOSTTEST_PATH] + test_objs + hip_objs + ["-lpthread"])
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
