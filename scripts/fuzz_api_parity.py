"""Seeded fuzz of the drop-in API against the reference built from /root/reference (oracle/_ref, test
infrastructure): ragged sizes, sample types, dark 16-bit content (mixed transfer-curve branches inside a wavefront),
tile modes, layouts.  tests/test_gpu_fuzz.py runs a few hundred cases of it in the suite; as a script it is the longer
sweep for spare GPU minutes.
usage: [FUZZ_BUDGET_S=seconds] python scripts/fuzz_api_parity.py [cases] [seed] [large]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

KINDS = ["photo", "smooth", "noise", "ramp", "black", "white"]


def sweep(cases, seed, large=False, lib=None, ref=None, budget_s=None):
    """Run `cases` seeded cases; returns (cases run, list of mismatch descriptions).  `large`: several LF groups per
    frame (2050-6200 px wide, up to 4200 high).  Stops early once `budget_s` seconds have passed."""
    from hydrium_amd import api, synth
    from oracle import refprobe

    if ref is None:
        refprobe.build()
        ref = refprobe.reference_library()
    lib = lib or api.Library()
    rng = np.random.default_rng(seed)
    bad = []
    t0 = time.time()
    done = 0
    for case in range(cases):
        if budget_s is not None and time.time() - t0 > budget_s:
            break
        w = int(rng.choice([rng.integers(1, 40), rng.integers(40, 300), rng.integers(300, 1100), rng.integers(2040, 2400)]))
        h = int(rng.choice([rng.integers(1, 40), rng.integers(40, 300), rng.integers(300, 1100)]))
        if large:
            w = int(rng.integers(2050, 6200))
            h = int(rng.choice([rng.integers(1, 300), rng.integers(2040, 4200)]))
        depth = int(rng.choice([8, 16, 16, 32]))
        kind = KINDS[int(rng.integers(len(KINDS)))]
        lin = int(rng.integers(4) == 0)
        if depth == 32:
            w, h = min(w, 500), min(h, 500)
            img = synth.make_image_f32(kind, w, h, seed=case + 1000 * seed)
        else:
            img = synth.make_image(kind, w, h, depth, seed=case + 1000 * seed)
            if depth == 16 and rng.integers(2):
                # dark content: part of the picture at or below the transfer curve's branch point (2650)
                img = (img >> int(rng.integers(1, 6))).astype(np.uint16)
                if rng.integers(2):
                    img[::3] = 0
            img = np.ascontiguousarray(img)
        kw = dict(linear_light=lin)
        mode = int(rng.integers(4))
        if mode == 1:
            kw.update(shift_x=int(rng.integers(0, 4)), shift_y=int(rng.integers(0, 4)))
        elif mode == 2:
            kw.update(layout="planar")
        elif mode == 3 and depth != 32:
            kw.update(layout="flipped")
        want = api.encode_image(ref, img, out_buf_size=1 << 22, **kw)
        got = api.encode_image(lib, img, out_buf_size=1 << 22, **kw)
        done += 1
        if got != want:
            bad.append(f"seed {seed} case {case}: {kind} {w}x{h} depth {depth} {kw} -> {len(got)} bytes, reference {len(want)}")
    return done, bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    sd = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t0 = time.time()
    budget = float(os.environ["FUZZ_BUDGET_S"]) if os.environ.get("FUZZ_BUDGET_S") else None
    ran, bad = sweep(n, sd, len(sys.argv) > 3 and sys.argv[3] == "large", budget_s=budget)
    for b in bad:
        print("MISMATCH", b, flush=True)
    print(f"{ran} cases, seed {sd}: {len(bad)} mismatches, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)
