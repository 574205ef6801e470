"""CPU-only: the oracle (and the oracle-fed host glue) against the committed golden fixtures, which
were produced from the real reference by tests/golden/make_golden.py.  This is what pins the
oracle on machines where /root/reference and oracle/_ref do not exist."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import binding as orc

import glue

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")

with open(os.path.join(GOLDEN, "manifest.json")) as _f:
    MANIFEST = json.load(_f)["files"]

CPU_BUDGET_PIXELS = 2200 * 2200  # keep the CPU suite to a few minutes


def test_stage_fixture_one_group(image):
    fx = np.load(os.path.join(GOLDEN, "stage_photo_256.npz"))
    res, _ = orc.encode_lf_group(image("photo", 256, 256, 8))
    ours = res.quant.copy()
    for c in range(3):
        ours[c, ::8, ::8] = res.dc[c]
    assert np.array_equal(fx["quant_with_lf"], ours)
    assert np.array_equal(fx["tokens"], res.symbols["token"])
    assert np.array_equal(fx["clusters"], res.symbols["cluster"])
    assert np.array_equal(fx["residue_bits"], res.symbols["residue_bits"])
    assert np.array_equal(fx["residues"], res.symbols["residue"])
    assert np.array_equal(fx["freqs"], res.freqs[:9])
    assert np.array_equal(fx["alphabet"], res.alphabet_size[:9])
    assert int(fx["section_bits"]) == res.group_bits[0]
    assert fx["section"].tobytes() == res.group_stream(0)


@pytest.mark.parametrize("entry", [e for e in MANIFEST if e["width"] * e["height"] <= CPU_BUDGET_PIXELS],
                         ids=lambda e: f"{e['kind']}-{e['width']}x{e['height']}-u{e['depth']}-s{e['shift']}")
def test_whole_files(image, entry):
    img = image(entry["kind"], entry["width"], entry["height"], entry["depth"])
    got = glue.encode_with_oracle_stages(img, entry["shift"], entry["shift"])
    assert len(got) == entry["size"]
    assert hashlib.md5(got).hexdigest() == entry["md5"]
    if "file" in entry:
        with open(os.path.join(GOLDEN, entry["file"]), "rb") as f:
            assert got == f.read()
