"""CPU-only: product host glue (frame.c / prefix.c / bitio.c via libhydrium_hosttest.so) fed with
oracle stages must reproduce the reference's whole file byte for byte."""
import hashlib

import numpy as np
import pytest

from hydrium_amd import api

import glue

ONE_FRAME = [
    ("photo", 256, 256, 8), ("photo", 8, 8, 8), ("ramp", 16, 16, 16), ("noise", 257, 255, 8), ("smooth", 1000, 700, 8),
    ("photo", 300, 200, 16), ("black", 64, 64, 8), ("white", 40, 520, 16), ("photo", 2049, 130, 8),
    ("smooth", 2100, 2060, 8),
]


@pytest.mark.parametrize("kind,w,h,depth", ONE_FRAME)
def test_one_frame_matches_reference(ref_lib, image, kind, w, h, depth):
    img = image(kind, w, h, depth)
    want = api.encode_image(ref_lib, img)
    got = glue.encode_with_oracle_stages(img)
    assert len(got) == len(want)
    assert got == want


@pytest.mark.parametrize("kind,w,h,depth", [("photo", 256, 256, 8), ("photo", 8, 8, 8), ("noise", 257, 255, 8),
                                           ("black", 64, 64, 8), ("white", 40, 520, 16), ("photo", 2049, 130, 8)])
def test_one_frame_from_coded_lf_streams(ref_lib, image, kind, w, h, depth):
    """hydamd_frame_from_streams: LF coefficient streams arrive already coded (what the GPU LF coder
    ships); the model of tests/lf_model.py stands in for the device here."""
    img = image(kind, w, h, depth)
    assert glue.encode_with_oracle_stages(img, coded_lf=True) == api.encode_image(ref_lib, img)


def test_tile_mode_from_coded_lf_streams(ref_lib, image):
    img = image("photo", 600, 520, 8)
    assert glue.encode_with_oracle_stages(img, 0, 0, coded_lf=True) == api.encode_image(ref_lib, img, shift_x=0, shift_y=0)


@pytest.mark.parametrize("shift", [0, 1, 2, 3])
def test_tile_mode_matches_reference(ref_lib, image, shift):
    img = image("photo", 1000, 700, 8)
    want = api.encode_image(ref_lib, img, shift_x=shift, shift_y=shift)
    got = glue.encode_with_oracle_stages(img, shift, shift)
    assert got == want


def test_mixed_tile_shifts(ref_lib, image):
    img = image("smooth", 700, 530, 16)
    want = api.encode_image(ref_lib, img, shift_x=1, shift_y=0)
    assert glue.encode_with_oracle_stages(img, 1, 0) == want


def test_out_of_order_tiles_permute_the_toc(ref_lib, image):
    img = image("photo", 2048 + 200, 2048 + 100, 8)
    order = [(1, 0), (0, 1), (0, 0), (1, 1)]
    want = api.encode_image(ref_lib, img, order=order)
    got = glue.encode_with_oracle_stages(img, order=order)
    assert got == want


def test_float_and_linear_light(ref_lib):
    from hydrium_amd import synth

    img = synth.make_image_f32("photo", 264, 136)
    assert glue.encode_with_oracle_stages(img) == api.encode_image(ref_lib, img)
    img16 = synth.make_image("photo", 120, 72, 16)
    assert glue.encode_with_oracle_stages(img16, linear_light=1) == api.encode_image(ref_lib, img16, linear_light=1)


def test_icc_profile_stream(ref_lib, image):
    rng = np.random.default_rng(7)
    head = bytearray(rng.integers(0, 256, 128, dtype=np.uint8).tobytes())
    head[36:40] = b"acsp"
    head[40:44] = b"APPL"
    body = b"".join(bytes([65 + (i * 7) % 26, 48 + i % 10, 0, 255][k % 4] for k in range(4)) for i in range(150))
    icc = bytes(head) + body
    img = image("photo", 72, 40, 8)
    want = api.encode_image(ref_lib, img, icc=icc)
    got = glue.encode_with_oracle_stages(img, icc=icc)
    assert got == want
    short = icc[:100]
    assert glue.encode_with_oracle_stages(img, icc=short) == api.encode_image(ref_lib, img, icc=short)


def test_survey_anchor_md5(image):
    """SURVEY.md Appendix C anchor that needs no reference build."""
    got = glue.encode_with_oracle_stages(image("photo", 256, 256, 8))
    assert (len(got), hashlib.md5(got).hexdigest()) == (12426, "469f79d37f5802edda18da52ff8dc889")
