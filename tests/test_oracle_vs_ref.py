"""Pin oracle/hyd_oracle.c against the real reference, stage by stage.

Runs only where oracle/_ref exists (the build container, or a GPU box that received the prebuilt
files).  The reference is driven through its public API with is_last=0 so that its intermediates
stay alive; oracle/ref_probe.c exposes them.
"""
import numpy as np
import pytest

from hydrium_amd import api
from oracle import binding as orc

CASES = [
    ("photo", 256, 256, 8),
    ("photo", 264, 200, 16),
    ("noise", 64, 48, 8),
    ("smooth", 517, 259, 8),
    ("photo", 97, 301, 16),
    ("ramp", 8, 8, 8),
    ("black", 40, 24, 8),
    ("white", 33, 9, 16),
]


def test_luts_match_reference(ref_probe, image):
    for depth, size in ((8, 256), (16, 65536)):
        with api.Encoder(ref_probe) as enc:
            enc.check(enc.set_metadata(16, 16))
            img = image("photo", 16, 16, depth)
            assert ref_probe.stage_xyb(enc, img) == 0
            getter = ref_probe.dll.refp_input_lut8 if depth == 8 else ref_probe.dll.refp_input_lut16
            ref_in = np.ctypeslib.as_array(getter(enc.h), (size,)).copy()
            ref_bias = np.ctypeslib.as_array(ref_probe.dll.refp_bias_lut(enc.h), (65536,)).copy()
        assert np.array_equal(ref_in, orc.input_lut(size, True))
        assert np.array_equal(ref_bias.view(np.uint32), orc.bias_lut().view(np.uint32))
    # linear-light LUT is the identity ramp through f32_to_u16
    with api.Encoder(ref_probe) as enc:
        enc.check(enc.set_metadata(16, 16, linear_light=1))
        assert ref_probe.stage_xyb(enc, image("photo", 16, 16, 16)) == 0
        ref_in = np.ctypeslib.as_array(ref_probe.dll.refp_input_lut16(enc.h), (65536,)).copy()
    assert np.array_equal(ref_in, orc.input_lut(65536, False))


@pytest.mark.parametrize("kind,w,h,depth", CASES)
def test_stages_match_reference(ref_probe, image, kind, w, h, depth):
    img = image(kind, w, h, depth)
    res, max_alpha = orc.encode_lf_group(img)

    # stage 1+2: XYB then DCT, bit-exact floats
    with api.Encoder(ref_probe) as enc:
        enc.check(enc.set_metadata(w, h))
        assert ref_probe.stage_xyb(enc, img) == 0
        xyb = ref_probe.xyb_planes(enc, 0, as_int=False)
        assert np.array_equal(xyb.view(np.uint32), res.xyb.view(np.uint32))
        ref_probe.stage_dct(enc, 0)
        dct = ref_probe.xyb_planes(enc, 0, as_int=False)
        # +0.0 vs -0.0 cannot influence any later stage (every consumer multiplies and truncates)
        assert np.array_equal(dct, res.dct) and not np.isnan(dct).any()

    # stages 3-6 through the public API with is_last = 0
    with api.Encoder(ref_probe) as enc:
        enc.check(enc.set_metadata(w, h))
        buf = api.C.create_string_buffer(1 << 16)
        enc.check(enc.provide_output_raw(api.C.cast(buf, api.C.c_void_p), len(buf)))
        enc.check(enc.send_tile(img, 0, 0, 2048, 2048, is_last=0))
        q = ref_probe.xyb_planes(enc, 0, as_int=True)
        ours = res.quant.copy()
        for c in range(3):
            ours[c, ::8, ::8] = res.dc[c]          # the reference keeps the LF int in the DC slot
        assert np.array_equal(q, ours)

        counts = ref_probe.group_symbol_counts(enc, 0, res.num_groups)
        assert np.array_equal(counts, res.group_symbols)
        syms = ref_probe.symbols(enc, int(counts.sum()))
        assert np.array_equal(syms, res.symbols)

        assert ref_probe.dll.refp_max_alphabet_size(enc.h) == res.max_alphabet_size == max_alpha
        for cl in range(res.cluster_from, res.cluster_to):
            n = ref_probe.dll.refp_alphabet_size(enc.h, cl)
            assert n == res.alphabet_size[cl]
            assert np.array_equal(ref_probe.frequencies(enc, cl), res.freqs[cl, :n])
        assert np.array_equal(ref_probe.cluster_map(enc), orc.hf_cluster_map(1))

        for g in range(res.num_groups):
            data, bits = ref_probe.group_stream(enc, g)
            assert bits == res.group_bits[g]
            assert data == res.group_stream(g)


def test_multi_lf_group_running_alphabet(ref_probe, image):
    """Two LF groups in one frame: presets, preset-id prefix bits and the running max alphabet."""
    img = image("photo", 2100, 300, 8)
    mx = 0
    with api.Encoder(ref_probe) as enc:
        enc.check(enc.set_metadata(2100, 300))
        buf = api.C.create_string_buffer(1 << 16)
        enc.check(enc.provide_output_raw(api.C.cast(buf, api.C.c_void_p), len(buf)))
        g0 = 0
        for tx in range(2):
            enc.check(enc.send_tile(img, tx, 0, 2048, 2048, is_last=0))
            res, mx = orc.encode_lf_group(img, tx, 0, max_alphabet_size=mx)
            counts = ref_probe.group_symbol_counts(enc, g0, g0 + res.num_groups)
            assert np.array_equal(counts, res.group_symbols)
            assert np.array_equal(ref_probe.symbols(enc, int(counts.sum())), res.symbols)
            for g in range(res.num_groups):
                data, bits = ref_probe.group_stream(enc, g0 + g)
                assert (data, bits) == (res.group_stream(g), res.group_bits[g])
            g0 += res.num_groups
        assert ref_probe.dll.refp_max_alphabet_size(enc.h) == mx


# 128 and 256 presets are absent on purpose: the reference never returns for them (its uint8_t loop
# counter in hyd_entropy_set_hybrid_config, entropy.c:99, cannot reach num_clusters == 256).
GRIDS = {1: (1, 1), 4: (2, 2), 28: (7, 4), 29: (29, 1), 85: (17, 5), 86: (43, 2), 127: (127, 1), 129: (43, 3),
         255: (17, 15)}


@pytest.mark.parametrize("num_presets", sorted(GRIDS))
def test_cluster_map_schemes(ref_probe, num_presets):
    """All four clustering schemes (encoder.c:862-901) against a reference frame with that many LF groups."""
    lfx, lfy = GRIDS[num_presets]
    w, h = 2048 * (lfx - 1) + 8, 2048 * (lfy - 1) + 8
    # The encoder is deliberately leaked: destroying the reference encoder mid-frame frees the
    # never-initialised writers of unsent groups (libhydrium.c:36-38 over encoder.c:914).
    enc = api.Encoder(ref_probe)
    enc.check(enc.set_metadata(w, h))
    buf = api.C.create_string_buffer(1 << 16)
    enc.check(enc.provide_output_raw(api.C.cast(buf, api.C.c_void_p), len(buf)))
    tile = np.zeros((8, 8, 3), np.uint8)
    ptr = tile.ctypes.data
    # lower-right tile is 8x8 pixels: cheap, and initialises the HF stream with the full cluster map
    enc.check(enc.send_tile_ptrs([ptr, ptr + 1, ptr + 2], lfx - 1, lfy - 1, 24, 3, 0, api.HYD_UINT8))
    assert np.array_equal(ref_probe.cluster_map(enc), orc.hf_cluster_map(num_presets))
    enc.h = None


def test_float_input_and_nan(ref_probe):
    from hydrium_amd import synth

    img = synth.make_image_f32("photo", 72, 40)
    res, _ = orc.encode_lf_group(img)
    with api.Encoder(ref_probe) as enc:
        enc.check(enc.set_metadata(72, 40))
        buf = api.C.create_string_buffer(1 << 16)
        enc.check(enc.provide_output_raw(api.C.cast(buf, api.C.c_void_p), len(buf)))
        enc.check(enc.send_tile(img, 0, 0, 2048, 2048, is_last=0))
        syms = ref_probe.symbols(enc, int(res.group_symbols.sum()))
        assert np.array_equal(syms, res.symbols)
        data, bits = ref_probe.group_stream(enc, 0)
        assert (data, bits) == (res.group_stream(0), res.group_bits[0])
    bad = img.copy()
    bad[3, 5, 1] = np.nan
    with pytest.raises(RuntimeError, match="-14"):
        orc.encode_lf_group(bad)
