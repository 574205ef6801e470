#!/usr/bin/env python3
"""A/B of build-time variants of the transform kernel (k_transform_tokenize) on one GPU box.

Every variant is the product library with kernels.hip recompiled under extra -D flags (HYDK_K1_GATHER,
HYDK_K1_WAVELOCAL, HYDK_K1_ILP, HYDK_K1_SKIP, ...: see the head of kernels.hip; the variants that lost were removed again, git history has them).  The run leg times the kernel ALONE
(one 8192x8192 RGB16 photo frame at a time, the library's own event timers) and hashes the frame's HF sections, so
a variant that changes a byte shows at once.  Variants alternate over the rounds, so box drift hits all alike.

    python scripts/k1_variants.py --build name=-DFLAG=1,-DOTHER=2 name2=...     # here (hipcc cross-compiles)
    python scripts/k1_variants.py --run [--rounds 3] [--pipe] name name2 ...    # on the GPU box
    python scripts/k1_variants.py --one name                                    # (internal) one measurement
"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "scripts", "probe_build")


def lib_of(name):
    return os.path.join(OUT, f"k1v_{name}.so")


def build(specs):
    from hydrium_amd import build as hb

    hb.build()
    os.makedirs(OUT, exist_ok=True)
    # the HYD_TEST_HOOKS flavour's objects (the probes skip stages through HYDAMD_DEBUG_SKIP), kernels.hip recompiled per variant
    names = sorted(os.listdir(hb.OBJ_DIR))
    base_objs = [os.path.join(hb.OBJ_DIR, f) for f in names
                 if f != "kernels.hip.o" and (f.endswith(".test.o") or (f.endswith(".o") and f[:-2] + ".test.o" not in names))]
    procs = []
    for spec in specs:
        name, _, flags = spec.partition("=")
        flags = [f for f in flags.split(",") if f]
        lf_only = [f[3:] for f in flags if f.startswith("LF:")]  # LF:<flag>: for lf_coder.hip alone (and that file is recompiled)
        flags = [f for f in flags if not f.startswith("LF:")]
        obj = os.path.join(OUT, f"k1v_{name}.kernels.o")
        cmd = [hb.HIPCC] + hb.HIP_FLAGS + flags + ["-c", os.path.join(hb.CSRC, "hip", "kernels.hip"), "-o", obj]
        pr2, obj2 = None, None
        if lf_only or any("HYDK_LF_" in f for f in flags):  # a switch lf_coder.hip reads too: that file is recompiled as well
            obj2 = os.path.join(OUT, f"k1v_{name}.lf_coder.o")
            pr2 = subprocess.Popen([hb.HIPCC] + hb.HIP_FLAGS + flags + lf_only + ["-c", os.path.join(hb.CSRC, "hip", "lf_coder.hip"), "-o", obj2],
                                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        procs.append((name, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True), obj2, pr2))
    for name, obj, pr, obj2, pr2 in procs:
        out, _ = pr.communicate()
        out2 = pr2.communicate()[0] if pr2 else ""
        if pr.returncode or (pr2 and pr2.returncode):
            print(f"{name}: COMPILE FAILED\n{out}{out2}")
            continue
        objs = [o for o in base_objs if not (obj2 and os.path.basename(o).startswith("lf_coder.hip"))] + [obj] + ([obj2] if obj2 else [])
        hb._run([hb.HIPCC, f"--offload-arch={hb.ARCH}", "-shared", "-fPIC", "-Wl,-soname,libhydrium.so.0", "-o", lib_of(name)]
                + objs + ["-lpthread"])
        os.remove(obj)
        if obj2:
            os.remove(obj2)
        print("built", lib_of(name))


def one(name, frames=6, pipe=False):
    os.environ["HYDAMD_LIB"] = lib_of(name)
    import torch

    from hydrium_amd import device, synth

    img = synth.make_image("photo", 8192, 8192, 16, device=torch.device("cuda", 0))
    res = {"name": name}
    with device.DeviceContext(0, 16, 0) as ctx:
        ctx.set_rans_waves(5)
        ctx.set_lf_coder(0)
        ctx.encode_image_tensor(img)
        ctx.sync()
        res["md5"] = hashlib.md5(ctx.read_payload()).hexdigest()[:12]
        ctx.profile(True)
        for _ in range(frames):
            ctx.encode_image_tensor(img)
            ctx.sync()
        prof = ctx.profile_read()
        ms, n = prof["transform_tokenize"]
        res["k1_ms"] = round(ms / max(n, 1), 4)
        ms, n = prof["rans_encode"]
        res["chain_ms"] = round(ms / max(n, 1), 4)
        ctx.profile(False)
        lds, regs = ctx.transform_footprint(1)
        res["lds"], res["vgpr"] = lds, regs
    if pipe:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "96", "--no-cpu-baseline", "--no-api", "--no-legs"],
                           capture_output=True, text=True, env=dict(os.environ))
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
            res["pipe_gpx"] = round(d["value"] / 1e3, 1)
        except Exception as e:  # noqa: BLE001
            res["pipe_gpx"] = f"failed: {e}: {r.stderr[-300:]}"
        if os.environ.get("K1V_NOISE"):  # the chain-bound loop: 2.9 symbols per pixel
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "48", "--kind", "noise", "--no-cpu-baseline", "--no-api",
                                "--no-legs", "--no-content"], capture_output=True, text=True, env=dict(os.environ))
            try:
                res["noise_gpx"] = round(json.loads(r.stdout.strip().splitlines()[-1])["value"] / 1e3, 1)
            except Exception as e:  # noqa: BLE001
                res["noise_gpx"] = f"failed: {e}: {r.stderr[-300:]}"
    print("RESULT " + json.dumps(res), flush=True)


def run(names, rounds, pipe):
    rows = {n: [] for n in names}
    for r in range(rounds):
        for n in names:
            cmd = [sys.executable, os.path.abspath(__file__), "--one", n] + (["--pipe"] if pipe else [])
            p = subprocess.run(cmd, capture_output=True, text=True)
            line = next((l for l in p.stdout.splitlines() if l.startswith("RESULT ")), None)
            if line is None:
                print(f"{n}: FAILED\n{p.stdout[-500:]}\n{p.stderr[-1500:]}", flush=True)
                continue
            d = json.loads(line[7:])
            rows[n].append(d)
            print(f"round {r} {n:14s} K1 {d['k1_ms']:.4f} ms  chains {d.get('chain_ms', 0):.4f} ms  md5 {d['md5']}  lds {d['lds']} vgpr {d['vgpr']}"
                  + (f"  pipelined {d['pipe_gpx']} Gpixel/s" if pipe else "") + (f"  noise loop {d['noise_gpx']}" if "noise_gpx" in d else ""), flush=True)
    print("\nsummary (K1 alone, ms: min / mean over rounds)")
    ref = rows[names[0]][0]["md5"] if rows[names[0]] else None
    for n in names:
        if not rows[n]:
            continue
        v = [d["k1_ms"] for d in rows[n]]
        c = [d.get("chain_ms", 0.0) for d in rows[n]]
        line = f"  {n:14s} {min(v):.4f} / {sum(v) / len(v):.4f}   chains {min(c):.4f}   bytes {'same' if rows[n][0]['md5'] == ref else 'DIFFER'}"
        if pipe:
            pv = [d["pipe_gpx"] for d in rows[n] if isinstance(d["pipe_gpx"], float)]
            if pv:
                line += f"   pipelined {min(pv):.1f} .. {max(pv):.1f} Gpixel/s"
        print(line)


if __name__ == "__main__":
    a = sys.argv[1:]
    if a and a[0] == "--build":
        build(a[1:])
    elif a and a[0] == "--one":
        one(a[1], pipe="--pipe" in a)
    elif a and a[0] == "--run":
        a = a[1:]
        rounds, pipe = 3, False
        if "--rounds" in a:
            i = a.index("--rounds")
            rounds = int(a[i + 1])
            del a[i:i + 2]
        if "--pipe" in a:
            pipe = True
            a.remove("--pipe")
        run(a, rounds, pipe)
    else:
        print(__doc__)
