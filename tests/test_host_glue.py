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


def test_repeated_lf_group_is_rejected_not_overrun(image):
    """the same LF group twice in a frame description used to write past the TOC arrays."""
    img = image("photo", 2100, 16, 8)
    with pytest.raises(RuntimeError, match="-14"):
        glue.encode_with_oracle_stages(img, order=[(0, 0), (0, 0)])
    with pytest.raises(RuntimeError, match="-14"):
        glue.encode_with_oracle_stages(img, order=[(1, 0), (1, 0)])


def _icc_unmangle(m: bytes) -> bytes:
    """Decoder side of the JPEG XL ICC transform for the streams hyd_set_suggested_icc_profile makes
    (header prediction, empty tag list, one 'copy the rest' command), written from the decoder's point
    of view: position i is predicted from bytes < i only."""
    def varint(buf, pos):
        v = shift = 0
        while True:
            b = buf[pos]
            pos += 1
            v |= (b & 127) << shift
            shift += 7
            if not b & 128:
                return v, pos
    size, pos = varint(m, 0)
    csize, pos = varint(m, pos)
    cmds = m[pos:pos + csize]
    data = m[pos + csize:]
    head = min(size, 128)
    out = bytearray()
    for i in range(head):
        p = 0
        if i < 4:
            p = (size >> (8 * (3 - i))) & 255
        elif i == 8:
            p = 4
        elif 12 <= i < 24:
            p = b"mntrRGB XYZ "[i - 12]
        elif 36 <= i < 40:
            p = b"acsp"[i - 36]
        elif 41 <= i < 44 and out[40] == ord("A"):
            p = b"APPL"[i - 40]
        elif 41 <= i < 44 and out[40] == ord("M"):
            p = b"MSFT"[i - 40]
        elif 42 <= i < 44 and out[40] == ord("S") and out[41] == ord("G"):
            p = b"SGI "[i - 40]
        elif 42 <= i < 44 and out[40] == ord("S") and out[41] == ord("U"):
            p = b"SUNW"[i - 40]
        elif i == 70:
            p = 246
        elif i == 71:
            p = 214
        elif i == 73:
            p = 1
        elif i == 78:
            p = 211
        elif i == 79:
            p = 45
        elif 80 <= i < 84:
            p = out[i - 76]
        out.append((data[i] + p) & 255)
    if size > 128:
        tags, cp = varint(cmds, 0)  # empty tag list
        assert tags == 0 and cmds[cp] == 1  # command 1: copy
        cnt, cp = varint(cmds, cp + 1)
        assert cnt == size - 128 and cp == len(cmds)
        out += data[head:head + cnt]
    return bytes(out)


@pytest.mark.parametrize("platform", [b"APPL", b"MSFT", b"SGI ", b"SUNW", b"SXYZ", b"\0\0\0\0"])
@pytest.mark.parametrize("size", [60, 128, 400])
def test_icc_transform_is_decodable_for_every_platform_signature(platform, size):
    """'SGI ' / 'SUNW' profiles crashed (the reference's "I "[i - 42] with i = 41)."""
    import ctypes as C

    rng = np.random.default_rng(11)
    icc = bytearray(rng.integers(0, 256, size, dtype=np.uint8).tobytes())
    icc[0:4] = size.to_bytes(4, "big")
    if size >= 44:
        icc[36:40] = b"acsp"
        icc[40:44] = platform
    if glue._d is None:
        glue._d = glue._lib()
    d = glue._d
    d.hydt_icc_mangled.restype = C.c_int
    d.hydt_icc_mangled.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    out, n = C.c_void_p(0), C.c_size_t(0)
    assert d.hydt_icc_mangled(bytes(icc), len(icc), C.byref(out), C.byref(n)) == 0
    mangled = bytes((C.c_uint8 * n.value).from_address(out.value))
    d.hydt_free(out)
    assert _icc_unmangle(mangled) == bytes(icc)


def _two_tiles_of_a_huge_image(lib, width, height, tiles, tile_imgs, shift=0):
    """Drive a libhydrium build in tile mode over an image far too large to hold: only the given tiles
    are sent (any tile but the last may be left out, libhydrium.h), each from its own small buffer."""
    import ctypes as C

    out = bytearray()
    with api.Encoder(lib) as enc:
        enc.check(enc.set_metadata(width, height, 0, shift, shift))
        buf = (C.c_uint8 * (1 << 20))()
        enc.check(enc.provide_output(buf))
        for i, ((tx, ty), img) in enumerate(zip(tiles, tile_imgs)):
            th, tw, _ = img.shape
            p = img.ctypes.data
            enc.check(enc.send_tile_ptrs([p, p + 1, p + 2], tx, ty, 3 * tw, 3, int(i == len(tiles) - 1), api.HYD_UINT8))
            while True:
                ret = enc.check(enc.flush())
                code, n = enc.release_output()
                enc.check(code)
                out += C.string_at(buf, n)
                enc.check(enc.provide_output(buf))
                if ret != api.HYD_NEED_MORE_OUTPUT:
                    break
    return bytes(out)


LEVEL10 = [((1 << 20) + 256, 256, [(0, 0), (4096, 0)]),        # wider than 2^20
           (20000, 20000, [(3, 5), (78, 78)])]                 # more than 2^28 pixels (78 = last 256-px tile, 32 px wide)


@pytest.mark.parametrize("width,height,tiles", LEVEL10)
def test_level10_container_prologue(ref_lib, image, width, height, tiles):
    """images beyond level 5 (a side over 2^20 or more than 2^28 pixels) start with the
    ISOBMFF signature + 'jxll' level box + an open-ended 'jxlc' box (reference encoder.c:23-30,170-174,
    libhydrium.c:67-68); nothing exercised that.  Host glue fed by the oracle against the reference."""
    from oracle import binding as orc

    imgs = []
    for k, (tx, ty) in enumerate(tiles):
        tw, th = min(256, width - tx * 256), min(256, height - ty * 256)
        imgs.append(np.ascontiguousarray(image("photo", tw, th, 8, seed=50 + k)))
    want = _two_tiles_of_a_huge_image(ref_lib, width, height, tiles, imgs)
    # signature box, 'ftyp', then the 'jxll' box saying level 10
    assert want[:12] == b"\0\0\0\x0cJXL \r\n\x87\n" and want[16:20] == b"ftyp" and b"jxll\x0a" in want[:64]
    md = api.HYDImageMetadata(width, height, 0, 0, 0)
    got = b""
    for k, ((tx, ty), img) in enumerate(zip(tiles, imgs)):
        r, mx = orc.encode_lf_group(img, num_presets=1, preset=0)
        got += glue.frame_from_stages(md, k == 0, k == len(tiles) - 1, [(tx, ty)], [r], mx)
    assert got == want


def test_frame_from_blobs_single_process_and_malformed_input(ref_lib, image):
    """hydamd_frame_from_blobs on the host: two shards' blobs (built from oracle results in the layout
    hydamd_export_frame writes) give the reference file; damaged, incomplete or repeated blobs give an
    error, not a crash."""
    import ctypes as C

    import torch

    import oracle_engine
    from hydrium_amd import device

    img = image("photo", 2048 + 300, 72, 8)
    if glue._d is None:
        glue._d = glue._lib()
    d = glue._d
    d.hydamd_frame_from_blobs.restype = C.c_int
    d.hydamd_frame_from_blobs.argtypes = [C.POINTER(api.HYDImageMetadata), C.c_int, C.c_int, C.c_size_t,
                                          C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t,
                                          C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_char_p)]
    blobs, floor = [], 0
    for lf in (0, 1):
        e = oracle_engine.OracleShardEngine(img, [lf])
        e.enqueue_transform()
        e.enqueue_entropy(torch.tensor([floor], dtype=torch.int32))
        out = torch.zeros(e.blob_bound(), dtype=torch.uint8)
        e.export_blob(out)
        n = int(device.blob_header(out[:64].numpy().tobytes())["total_bytes"])
        blobs.append(out[:n].numpy().tobytes())
        floor = max(floor, e.results[-1][2])
    md = api.HYDImageMetadata(img.shape[1], img.shape[0], 0, -1, -1)
    assert device.frame_from_blobs(md, blobs, lib=d) == api.encode_image(ref_lib, img)

    def fails(bl):
        with pytest.raises(device.DeviceError):
            device.frame_from_blobs(md, bl, lib=d)

    fails([blobs[0]])                                   # an LF group is missing
    fails([blobs[0], blobs[0]])                         # an LF group twice
    fails([blobs[0], blobs[1][:100]])                   # truncated
    fails([blobs[0], b"\0" * len(blobs[1])])            # no magic
    bad = bytearray(blobs[1])
    bad[12] |= 2                                        # header.status: "this shard must rerun its frame"
    fails([blobs[0], bytes(bad)])
    huge = bytearray(blobs[1])
    huge[64 + 16 + 36 + 12 + 256 + 4608 + 16:64 + 16 + 36 + 12 + 256 + 4608 + 20] = (1 << 30).to_bytes(4, "little")  # lf.offset beyond the blob
    fails([blobs[0], bytes(huge)])


def test_frame_buffers_are_reused_without_changing_a_byte(ref_lib, image):
    """Frames above 1 MB leave through a buffer the library may keep when it comes back (hydamd_free) and hand to the
    next frame: a larger frame, a smaller one and the first again give the reference's bytes each time, before and
    after hydamd_trim_cache."""
    big = image("noise", 1024, 768, 8)       # about 1.4 MB of codestream
    small = image("noise", 640, 512, 8)
    tiny = image("photo", 300, 200, 8)       # below the size the library keeps
    want = {id(a): api.encode_image(ref_lib, a, out_buf_size=1 << 22) for a in (big, small, tiny)}
    assert len(want[id(big)]) > 1 << 20
    for a in (big, small, tiny, big, small):
        assert glue.encode_with_oracle_stages(a) == want[id(a)]
    glue._lib().hydamd_trim_cache()
    for a in (small, big):
        assert glue.encode_with_oracle_stages(a) == want[id(a)]
