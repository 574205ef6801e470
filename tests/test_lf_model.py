"""CPU: the host splice around a device-style LF-coefficient stream (hyd_write_lf_group_coded) must
reproduce the host-coded LFGroup section bit for bit.  The stream here comes from tests/lf_model.py,
the numpy model the GPU tests hold the device to."""
import numpy as np
import pytest

from tests import lf_model


def _dc(kind, vbw, vbh, seed=0):
    rng = np.random.default_rng(seed)
    if kind == "zeros":
        return np.zeros((3, vbh, vbw), np.int32)
    if kind == "const":
        return np.full((3, vbh, vbw), 977, np.int32)
    if kind == "smooth":
        y, x = np.mgrid[0:vbh, 0:vbw]
        return np.stack([(x * 3 + y * 5) // 4, (x * x + y) // 7, (x + y * y) // 9]).astype(np.int32)
    if kind == "noise":
        return rng.integers(-3000, 3000, (3, vbh, vbw)).astype(np.int32)
    if kind == "huge":  # large residuals: long residue fields
        return rng.integers(-(1 << 29), 1 << 29, (3, vbh, vbw)).astype(np.int32)
    if kind == "runs":  # long runs with interruptions at awkward places (chunk and min-length edges)
        a = np.zeros(3 * vbh * vbw, np.int32)
        pos = 0
        for ln in [1, 2, 3, 4, 5, 6, 127, 128, 129, 130, 131, 132, 133, 255, 256, 257, 260, 400, 3, 700]:
            pos += ln
            if pos < len(a):
                a[pos:] += 1 + (ln % 3)
        return a.reshape(3, vbh, vbw)
    raise ValueError(kind)


CASES = [("zeros", 1, 1), ("zeros", 32, 32), ("const", 17, 9), ("smooth", 32, 32), ("smooth", 100, 37),
         ("noise", 32, 32), ("noise", 5, 3), ("huge", 16, 16), ("runs", 64, 40), ("runs", 128, 3)]


@pytest.mark.parametrize("kind,vbw,vbh", CASES)
def test_coded_splice_matches_host_coder(kind, vbw, vbh):
    dc = _dc(kind, vbw, vbh)
    want = lf_model.host_lf_group(dc)
    hist, lengths, alphabet, pairs, bits, nbits = lf_model.model(dc)
    got = lf_model.coded_lf_group(vbw, vbh, lengths, alphabet, pairs, bits, nbits)
    assert got == want


def test_model_run_rules():
    # 1 literal + r repeats: r <= 3 -> literals, r > 3 -> one pair; chunks of 128
    v = np.array([5] * 4 + [6] * 5 + [7] * 128 + [8] * 129 + [9] * 133, np.uint64)
    lit, r = lf_model.emissions(v)
    assert lit[:4].all() and not r[:4].any()                      # 5 x4: literal + 3 literals
    assert lit[4] and r[4] == 4 and not lit[5:9].any()            # 6 x5: literal + run of 4
    assert lit[9] and r[9] == 127 and not lit[10:137].any()       # 7 x128: exactly one chunk
    s = 137
    assert lit[s] and r[s] == 127 and lit[s + 128] and r[s + 128] == 0  # 8 x129: chunk + lone literal
    s = 137 + 129
    assert r[s] == 127 and lit[s + 128] and r[s + 128] == 4       # 9 x133: chunk + literal + run of 4
