"""GPU: the LF-group coder (csrc/hip/lf_coder.hip) against the host prefix coder and the numpy
model of tests/lf_model.py — code lengths, canonical codes, symbol bits and whole LFGroup sections."""
import numpy as np
import pytest

from hydrium_amd import device as dev, synth
from tests import lf_model

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    with dev.DeviceContext(0, 4) as c:
        yield c


def _hists():
    rng = np.random.default_rng(7)
    out = []

    def h(lit=None, run=None):
        a = np.zeros(lf_model.LF_CODES, np.uint32)
        if lit is not None:
            a[: len(lit)] = lit
        if run is not None:
            a[257: 257 + len(run)] = run
        return a

    out.append(("single", h([0, 0, 9])))
    out.append(("two", h([4, 0, 9])))
    out.append(("three-equal", h([5, 5, 5])))
    out.append(("four-skewed", h([100, 20, 3, 1])))
    out.append(("flat-16", h([7] * 16)))
    out.append(("flat-17", h([7] * 17)))
    out.append(("flat-228", h([1] * 228)))
    out.append(("flat-all", h([3] * 228, [3] * 124)))
    out.append(("ones-and-runs", h([1] * 40, [1] * 124)))
    out.append(("fibonacci", h([1, 1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 144, 233, 377, 610, 987, 1597, 2584, 4181, 6765])))
    out.append(("fibonacci-long", h([int(1.618 ** i) + 1 for i in range(30)])))   # forces the depth limit
    out.append(("geometric", h([1 << i for i in range(20)], [1 << (i % 17) for i in range(60)])))
    out.append(("runs-only-top", h([50], [0] * 123 + [2])))
    for i in range(40):
        nl, nr = int(rng.integers(1, 229)), int(rng.integers(0, 125))
        style = i % 4
        if style == 0:    # many ties
            lit, run = rng.integers(0, 4, nl), rng.integers(0, 3, nr)
        elif style == 1:  # photo-like decay
            lit = (20000 * np.exp(-np.arange(nl) / 9.0)).astype(np.int64) + rng.integers(0, 3, nl)
            run = rng.integers(0, 50, nr)
        elif style == 2:  # heavy tail that hits the depth limit
            lit = np.maximum(1, (3.0 ** rng.integers(0, 12, nl))).astype(np.int64)
            run = rng.integers(0, 2, nr)
        else:
            lit, run = rng.integers(0, 100000, nl), rng.integers(0, 1000, nr)
        a = h(lit.astype(np.uint32), run.astype(np.uint32))
        if not a.any():
            a[3] = 1
        out.append((f"random-{i}", a))
    # the coder's weights are bounded by 2 symbols per LF value (393216 per LF group); stay below 2^20
    fitted = []
    for name, a in out:
        while int(a.sum()) >= MAX_TOTAL:
            a = np.where(a > 0, np.maximum(1, a // 2), 0).astype(np.uint32)
        fitted.append((name, a))
    return fitted


MAX_TOTAL = 1 << 20


def test_code_construction_rejects_overweight(ctx):
    hist = np.zeros(lf_model.LF_CODES, np.uint32)
    hist[:4] = MAX_TOTAL // 4
    _, _, _, err = ctx.debug_lf_code(hist)
    assert err != 0


@pytest.mark.parametrize("name,hist", _hists(), ids=[n for n, _ in _hists()])
def test_code_construction_matches_host(ctx, name, hist):
    lengths, codes, alphabet, err = ctx.debug_lf_code(hist)
    assert err == 0
    _, want_alpha = lf_model.compact_to_tokens(hist)
    assert alphabet == want_alpha
    want = lf_model.lengths_for_hist(hist)
    np.testing.assert_array_equal(lengths, want)
    np.testing.assert_array_equal(codes, lf_model.canonical_codes(want))


IMAGES = [("photo", 256, 256, 8), ("photo", 777, 513, 16), ("smooth", 1024, 640, 8), ("noise", 300, 200, 8),
          ("black", 2048, 2048, 8), ("white", 1031, 9, 16), ("ramp", 2048, 1111, 8), ("photo", 2048, 2048, 16),
          ("smooth", 8, 8, 8)]


@pytest.mark.parametrize("kind,w,h,depth", IMAGES)
def test_lf_stream_matches_model(kind, w, h, depth):
    import torch

    img = synth.make_image(kind, w, h, depth, seed=3)
    t = torch.from_numpy(img).cuda()
    with dev.DeviceContext(0, 1) as c:
        assert c.lf_coder()
        c.encode_image_tensor(t)
        c.sync()
        vbw, vbh = (w + 7) // 8, (h + 7) // 8
        dc = c.read_dc(0, vbw, vbh)
        lengths, alphabet, pairs, nbits = c.read_lf_stream(0)
        bits = c.read_lf_bits(0, nbits)
    hist, m_len, m_alpha, m_pairs, m_bits, m_nbits = lf_model.model(dc)
    assert (alphabet, pairs, nbits) == (m_alpha, m_pairs, m_nbits)
    np.testing.assert_array_equal(lengths, m_len)
    # bits past bit_count in the last byte are unspecified on the device side
    if nbits % 8:
        mask = (1 << (nbits % 8)) - 1
        assert (int(bits[-1]) & mask) == (int(m_bits[-1]) & mask)
        np.testing.assert_array_equal(bits[:-1], m_bits[:-1])
    else:
        np.testing.assert_array_equal(bits, m_bits)
    # and the whole LFGroup section equals the host coder's
    assert lf_model.coded_lf_group(vbw, vbh, lengths, alphabet, pairs, bits, nbits) == lf_model.host_lf_group(dc)


def test_float_input_large_residuals():
    import torch

    img = synth.make_image_f32("noise", 520, 264, seed=5) * 40.0 - 3.0   # far outside [0,1]: long residues
    t = torch.from_numpy(img).cuda()
    with dev.DeviceContext(0, 1) as c:
        c.encode_image_tensor(t)
        c.sync()
        dc = c.read_dc(0, 65, 33)
        lengths, alphabet, pairs, nbits = c.read_lf_stream(0)
        bits = c.read_lf_bits(0, nbits)
    assert lf_model.coded_lf_group(65, 33, lengths, alphabet, pairs, bits, nbits) == lf_model.host_lf_group(dc)


def test_api_bytes_identical_with_host_lf_coder(monkeypatch):
    from hydrium_amd import api

    img = synth.make_image("photo", 2300, 2100, 8, seed=11)   # 2 x 2 LF groups, ragged
    lib = api.Library()
    monkeypatch.setenv("HYDAMD_LF_CODER", "1")
    a = api.encode_image(lib, img)
    monkeypatch.setenv("HYDAMD_LF_CODER", "0")
    b = api.encode_image(lib, img)
    assert a == b


def test_packed_lf_payload_matches_per_slot_reads():
    import torch

    img = synth.make_image("photo", 2048 + 520, 2048 + 264, 8, seed=9)   # 2 x 2 LF groups of four different shapes
    t = torch.from_numpy(img).cuda()
    with dev.DeviceContext(0, 4) as c:
        c.encode_image_tensor(t)
        c.sync()
        info = c.read_lf_streams(4)
        blob = c.read_lf_payload()
        on_dev = c.lf_payload_tensor().cpu().numpy()
        np.testing.assert_array_equal(on_dev, blob)
        end = 0
        for s in range(4):
            lengths, alphabet, pairs, nbits = c.read_lf_stream(s)
            assert (int(info["bit_count"][s]), int(info["alphabet"][s]), int(info["run_pairs"][s])) == (nbits, alphabet, pairs)
            np.testing.assert_array_equal(info["lengths"][s], lengths)
            off = int(info["offset"][s])
            assert off == end and off % 4 == 0
            nbytes = (nbits + 7) // 8
            np.testing.assert_array_equal(blob[off:off + nbytes], c.read_lf_bits(s, nbits))
            end = off + (nbits + 31) // 32 * 4
        assert end == len(blob)


def test_early_lf_streams_equal_the_batched_ones():
    """hyd_send_tile's schedule at the device level: transform stages tile by tile, the LF coder in
    two instalments in front of the entropy stage, its streams read after sync_lf() while the entropy
    stage is still queued — same LF streams and same HF sections as the one-shot finish_frame."""
    import torch

    img = synth.make_image("photo", 2048 + 520, 2048 + 264, 16, seed=21)   # 2 x 2 LF groups
    t = torch.from_numpy(img.view(np.int16)).cuda()
    with dev.DeviceContext(0, 4) as c:
        c.encode_image_tensor(t)
        c.sync()
        want_info, want_blob, want_payload = c.read_lf_streams(4), c.read_lf_payload(), c.read_payload()

        h, w, _ = img.shape
        c.begin_frame(4)
        for slot in range(4):
            x0, y0 = (slot % 2) * 2048, (slot // 2) * 2048
            p = t.data_ptr() + (y0 * w + x0) * 3 * 2
            c.encode_lf_group(slot, [p, p + 2, p + 4], 3 * w, 3, 1, min(2048, w - x0), min(2048, h - y0), slot)
            c.submit_lf_group(slot)
            if slot == 1:
                c.run_lf_coder(2, False)
        c.run_lf_coder(4, True)
        c.finish_frame(4)
        c.sync_lf()
        info, blob = c.read_lf_streams(4), c.read_lf_payload()
        c.sync()
        assert c.read_payload() == want_payload
    for k in ("bit_count", "alphabet", "run_pairs", "offset", "lengths"):
        np.testing.assert_array_equal(info[k], want_info[k])
    np.testing.assert_array_equal(blob, want_blob)
