"""Generate the golden fixtures from the REAL reference (oracle/_ref, needs /root/reference).

Run in the build container:  python tests/golden/make_golden.py
Writes small whole codestreams as files, an MD5 + size manifest for larger ones, and stage-level
fixtures (tokens, frequencies, LF ints, section bytes of one 256x256 group) as .npz.  Fixtures are
data only: inputs are regenerated from hydrium_amd/synth.py by name.
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from hydrium_amd import api, synth  # noqa: E402
from oracle import refprobe  # noqa: E402

SMALL = [  # committed as files
    ("photo", 256, 256, 8, -1), ("photo", 8, 8, 8, -1), ("ramp", 16, 16, 16, -1), ("noise", 64, 48, 8, -1),
    ("smooth", 257, 255, 8, -1), ("photo", 200, 120, 16, -1), ("photo", 300, 280, 8, 0),
]
LARGE = [  # pinned by MD5 + size
    ("photo", 1000, 700, 8, -1), ("photo", 1000, 700, 8, 0), ("photo", 1000, 700, 8, 1), ("photo", 2048, 2048, 8, -1),
    ("photo", 2048, 2048, 16, -1), ("smooth", 2100, 2060, 8, -1), ("photo", 4096, 4096, 8, -1),
    ("smooth", 4096, 4096, 8, -1), ("noise", 2048, 2048, 8, -1), ("photo", 3840, 2160, 8, -1),
]


def main():
    assert refprobe.build(), "reference build unavailable"
    lib = refprobe.reference_library()
    files = []
    for kind, w, h, depth, shift in SMALL + LARGE:
        img = synth.make_image(kind, w, h, depth)
        data = api.encode_image(lib, img, shift_x=shift, shift_y=shift)
        entry = dict(kind=kind, width=w, height=h, depth=depth, shift=shift, size=len(data),
                     md5=hashlib.md5(data).hexdigest())
        if (kind, w, h, depth, shift) in SMALL:
            name = f"{kind}_{w}x{h}_u{depth}_s{shift}.jxl"
            with open(os.path.join(HERE, name), "wb") as f:
                f.write(data)
            entry["file"] = name
        files.append(entry)
        print(entry)
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_golden.py", "reference": "Traneptora/hydrium v0.6.0 (gcc -Os)",
                   "seed": 1234, "files": files}, f, indent=1)

    # stage-level fixture: one 256x256 photo group through the reference's own internals
    probe = refprobe.Probe()
    img = synth.make_image("photo", 256, 256, 8)
    with api.Encoder(probe) as enc:
        enc.check(enc.set_metadata(256, 256))
        buf = api.C.create_string_buffer(1 << 16)
        enc.check(enc.provide_output_raw(api.C.cast(buf, api.C.c_void_p), len(buf)))
        enc.check(enc.send_tile(img, 0, 0, 2048, 2048, is_last=0))
        q = probe.xyb_planes(enc, 0, as_int=True)
        n = int(probe.group_symbol_counts(enc, 0, 1)[0])
        syms = probe.symbols(enc, n)
        freqs = np.zeros((9, 128), np.uint32)
        alpha = np.zeros(9, np.uint32)
        for c in range(9):
            f = probe.frequencies(enc, c)
            freqs[c, :len(f)] = f
            alpha[c] = len(f)
        stream, bits = probe.group_stream(enc, 0)
    np.savez_compressed(os.path.join(HERE, "stage_photo_256.npz"), quant_with_lf=q, tokens=syms["token"],
                        clusters=syms["cluster"], residue_bits=syms["residue_bits"], residues=syms["residue"],
                        freqs=freqs, alphabet=alpha, section=np.frombuffer(stream, np.uint8), section_bits=bits)
    print("stage fixture:", n, "symbols,", bits, "bits")


if __name__ == "__main__":
    main()
