"""GPU, through the drop-in C API (hyd_encoder_new ... hyd_send_tile ... hyd_flush): the bytes that
come out of hydrium_amd/lib/libhydrium.so.0 must equal the reference's.

Three anchors, strongest available first:
  * oracle/_ref/libhydrium_ref.so (the real reference, when the prebuilt file travelled),
  * the committed golden fixtures / MD5 manifest under tests/golden (generated from the reference),
  * the host glue fed by the CPU oracle (tests/glue.py), itself pinned to the reference on CPU.
"""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import has_gpu, reference_expected
from hydrium_amd import api

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def lib():
    return api.Library()


def _expected(img, **kw):
    """The bytes to match, from the anchor conftest.parity_anchor() names — never a silent fallback: without
    oracle/_ref (and without HYDAMD_ALLOW_ORACLE_ANCHOR=1) every test that asks fails."""
    from conftest import parity_anchor
    from oracle import refprobe

    anchor = parity_anchor()
    if anchor == "reference":
        return api.encode_image(refprobe.reference_library(), img, **kw), anchor
    assert anchor == "oracle+glue", ("oracle/_ref (the compiled reference) did not travel with this tree: build it with "
                                     "`make -C oracle ref` where /root/reference exists, or accept the weaker anchor "
                                     "explicitly with HYDAMD_ALLOW_ORACLE_ANCHOR=1")
    import glue

    g = {k: v for k, v in kw.items() if k in ("shift_x", "shift_y", "order", "icc", "linear_light")}
    return glue.encode_with_oracle_stages(img, **g), anchor


def _anchor_id():
    from conftest import parity_anchor

    return parity_anchor()


@pytest.mark.parametrize("anchor", [_anchor_id()])
def test_parity_anchor(anchor):
    """The anchor every whole-file comparison of this module used, spelled out in the test id."""
    assert anchor in ("reference", "oracle+glue"), "no parity anchor: oracle/_ref is absent (see _expected)"


CASES = [
    ("photo", 256, 256, 8), ("photo", 8, 8, 8), ("noise", 257, 255, 8), ("smooth", 1000, 700, 8),
    ("photo", 300, 200, 16), ("black", 64, 64, 8), ("white", 40, 520, 16), ("photo", 2049, 130, 8),
    ("smooth", 2100, 2060, 8), ("photo", 4096, 4096, 8),
]


@pytest.mark.parametrize("kind,w,h,depth", CASES)
def test_one_frame_files(lib, image, kind, w, h, depth):
    img = image(kind, w, h, depth)
    want, _ = _expected(img)
    got = api.encode_image(lib, img)
    assert len(got) == len(want)
    assert got == want


@pytest.mark.parametrize("layout", ["planar", "flipped"])
def test_layouts(lib, image, layout):
    img = image("photo", 520, 300, 8)
    want, _ = _expected(img)
    assert api.encode_image(lib, img, layout=layout) == want


@pytest.mark.parametrize("shift", [0, 1, 3])
def test_tile_mode(lib, image, shift):
    img = image("photo", 1000, 700, 8)
    want, _ = _expected(img, shift_x=shift, shift_y=shift)
    assert api.encode_image(lib, img, shift_x=shift, shift_y=shift) == want


def test_out_of_order_tiles(lib, image):
    img = image("photo", 2048 + 200, 2048 + 100, 8)
    order = [(1, 0), (0, 1), (0, 0), (1, 1)]
    want, _ = _expected(img, order=order)
    assert api.encode_image(lib, img, order=order) == want


def test_tiny_output_buffer_streams_the_same_bytes(lib, image):
    img = image("photo", 300, 200, 8)
    want, _ = _expected(img)
    assert api.encode_image(lib, img, out_buf_size=64) == want
    assert api.encode_image(lib, img, out_buf_size=4099) == want


def test_float_linear_and_icc(lib, image):
    from hydrium_amd import synth

    f = synth.make_image_f32("photo", 264, 136)
    assert api.encode_image(lib, f) == _expected(f)[0]
    u16 = image("photo", 120, 72, 16)
    assert api.encode_image(lib, u16, linear_light=1) == _expected(u16, linear_light=1)[0]
    icc = bytes(range(256)) * 2
    img = image("photo", 72, 40, 8)
    assert api.encode_image(lib, img, icc=icc) == _expected(img, icc=icc)[0]


def test_nan_sample_is_an_api_error(lib):
    img = np.zeros((16, 16, 3), np.float32)
    img[3, 3, 0] = np.nan
    with pytest.raises(api.HydriumError) as ei:
        api.encode_image(lib, img)
    assert ei.value.code == api.HYD_API_ERROR and ei.value.message == "Invalid NaN Float"


def test_nan_sample_in_a_pipelined_tile_frame_is_an_api_error(lib, image):
    """Tile mode keeps frames in flight: the error of a tile's pixels surfaces when its frame is collected — a later
    call, at the latest the final tile's — with the reference's message; the library is usable afterwards."""
    from hydrium_amd import synth

    f = synth.make_image_f32("photo", 600, 520).copy()
    f[300, 300, 1] = np.nan                                # tile (1, 1) of 3 x 3
    with pytest.raises(api.HydriumError) as ei:
        api.encode_image(lib, f, shift_x=0, shift_y=0)
    assert ei.value.code == api.HYD_API_ERROR and ei.value.message == "Invalid NaN Float"
    good = image("photo", 600, 520, 8)
    assert api.encode_image(lib, good, shift_x=0, shift_y=0) == _expected(good, shift_x=0, shift_y=0)[0]


def test_golden_manifest(lib, image):
    """Fixtures generated from the reference by tests/golden/make_golden.py."""
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        manifest = json.load(f)
    for entry in manifest["files"]:
        if entry["width"] * entry["height"] > 4096 * 4096:
            continue
        img = image(entry["kind"], entry["width"], entry["height"], entry["depth"])
        got = api.encode_image(lib, img, shift_x=entry["shift"], shift_y=entry["shift"])
        assert len(got) == entry["size"], entry
        assert hashlib.md5(got).hexdigest() == entry["md5"], entry
        if "file" in entry:
            with open(os.path.join(GOLDEN, entry["file"]), "rb") as fh:
                assert got == fh.read()


def test_abandoned_frame_leaves_a_reusable_context(lib, image):
    """An encoder destroyed half-way through a frame parks its device context (encoder.c context
    reuse); the next encoder of the same shape must start from a clean slate."""
    img = image("photo", 2048 + 300, 2048 + 200, 8)     # 2 x 2 LF groups
    want, _ = _expected(img)
    enc = api.Encoder(lib)
    enc.check(enc.set_metadata(img.shape[1], img.shape[0]))
    import ctypes as C
    buf = (C.c_uint8 * (1 << 20))()
    enc.check(enc.provide_output(buf))
    enc.check(enc.send_tile(img, 0, 0, 2048, 2048))
    enc.check(enc.send_tile(img, 1, 0, 2048, 2048))
    enc.close()                                          # two of four tiles sent, kernels queued
    other = image("noise", 2048 + 300, 2048 + 200, 8)    # same shape, different content in between
    assert api.encode_image(lib, other) == _expected(other)[0]
    assert api.encode_image(lib, img) == want


@pytest.mark.parametrize("shift,w,h,depth", [(0, 1300, 1100, 8), (1, 2300, 1500, 16), (0, 700, 2400, 16)])
def test_tile_mode_with_more_tiles_than_frames_in_flight(lib, image, shift, w, h, depth):
    """Tile-mode frames are pipelined (encoder.c tile_pipeline_depth: up to eight in flight, each on a device context of
    its own, collected in send order): images of 30, 15 and 30 tiles, ragged at the right and bottom edges — the bytes
    are the reference's, through the CLI's call pattern (flush after every tile) and the documented one."""
    img = image("photo", w, h, depth)
    want, _ = _expected(img, shift_x=shift, shift_y=shift)
    for ring in (8, 3):
        assert api.encode_image(lib, img, shift_x=shift, shift_y=shift, tile_pipeline=ring) == want
        assert _encode_documented_protocol(lib, img, 1 << 16, shift, tile_pipeline=ring) == want
    assert api.encode_image(lib, img, shift_x=shift, shift_y=shift) == want  # the default: one frame per call


def _tile_calls(lib, img, shift, tiles, pipeline=None):
    """Bytes handed over after each hyd_send_tile + flush loop, call by call (the reference CLI's loop)."""
    import ctypes as C

    h, w, _ = img.shape
    tw, th = api.tile_dims(w, h, shift, shift)
    per_call = []
    with api.Encoder(lib) as enc:
        enc.check(enc.set_metadata(w, h, 0, shift, shift))
        if pipeline is not None:
            enc.check(enc.set_tile_pipeline(pipeline))
        buf = (C.c_uint8 * (1 << 20))()
        enc.check(enc.provide_output(buf))
        for tx, ty in tiles:
            enc.check(enc.send_tile(img, tx, ty, tw, th))
            got = bytearray()
            while True:
                ret = enc.check(enc.flush())
                code, n = enc.release_output()
                enc.check(code)
                got += C.string_at(buf, n)
                enc.check(enc.provide_output(buf))
                if ret != api.HYD_NEED_MORE_OUTPUT:
                    break
            per_call.append(bytes(got))
    return per_call  # the encoder is destroyed here, possibly mid-image


def test_tile_mode_default_timing_is_the_reference_s(lib, image):
    """SURVEY 8(b) "output timing": by default every hyd_send_tile call completes a whole frame — after the call's own
    flush loop the caller holds exactly the bytes the reference has handed over after the same call, tile by tile; an
    encoder destroyed mid-image has therefore emitted every frame it collected (reference libhydrium.c:147-203)."""
    from oracle import refprobe

    assert reference_expected(), "this test compares call by call with the compiled reference"
    img = image("photo", 1300, 1100, 8)
    tiles = [(t % 6, t // 6) for t in range(11)]          # 11 of 30 tiles, then the encoder is destroyed
    want = _tile_calls(refprobe.reference_library(), img, 0, tiles)
    with api.Encoder(lib) as enc:
        assert enc.tile_pipeline() == int(os.environ.get("HYDAMD_TILE_PIPELINE", "1"))
    got = _tile_calls(lib, img, 0, tiles, pipeline=1)
    assert [len(g) for g in got] == [len(x) for x in want]
    assert got == want
    assert all(len(g) > 0 for g in got)
    # opted into a ring of four: the same bytes in the same order, later — and the abandoned image loses what was in flight
    ring = _tile_calls(lib, img, 0, tiles, pipeline=4)
    assert b"".join(ring) == b"".join(want[:len(tiles) - 4])  # call k collected frame k - 4; four frames were dropped
    assert [len(r) for r in ring[1:4]] == [0, 0, 0] and all(len(r) > 0 for r in ring[4:])


def test_tile_pipeline_knob(lib, image):
    import ctypes as C

    img = image("photo", 600, 300, 8)
    enc = api.Encoder(lib)
    enc.check(enc.set_metadata(600, 300, 0, 0, 0))
    assert enc.set_tile_pipeline(9) == api.HYD_API_ERROR and enc.set_tile_pipeline(-1) == api.HYD_API_ERROR
    enc.check(enc.set_tile_pipeline(4))
    assert enc.tile_pipeline() == 4
    buf = (C.c_uint8 * (1 << 20))()
    enc.check(enc.provide_output(buf))
    enc.check(enc.send_tile(img, 0, 0, 256, 256))
    assert enc.set_tile_pipeline(1) == api.HYD_API_ERROR   # a frame is in flight
    assert "in flight" in enc.error_message()
    enc.close()


def test_tile_mode_image_abandoned_with_frames_in_flight(lib, image):
    """An encoder destroyed (or given new metadata) while tile frames are in flight drops them and hands clean contexts
    back; what follows — the same encoder on a new image, then a fresh one — is unaffected."""
    import ctypes as C

    img = image("photo", 1300, 1100, 8)
    want, _ = _expected(img, shift_x=0, shift_y=0)
    enc = api.Encoder(lib)
    enc.check(enc.set_metadata(1300, 1100, 0, 0, 0))
    enc.check(enc.set_tile_pipeline(8))
    buf = (C.c_uint8 * (1 << 20))()
    enc.check(enc.provide_output(buf))
    for t in range(11):                                   # more than one lap of the ring, none of them the last tile
        enc.check(enc.send_tile(img, t % 6, t // 6, 256, 256))
    other = image("smooth", 600, 520, 8)                  # new metadata on the same encoder: frames in flight are dropped
    enc.check(enc.set_metadata(600, 520, 0, 1, 1))
    enc.release_output()
    enc.check(enc.provide_output(buf))
    out = bytearray()
    for ty in range(2):
        for tx in range(2):
            enc.check(enc.send_tile(other, tx, ty, 512, 512))
            while True:
                ret = enc.check(enc.flush())
                code, n = enc.release_output()
                enc.check(code)
                out += C.string_at(buf, n)
                enc.check(enc.provide_output(buf))
                if ret != api.HYD_NEED_MORE_OUTPUT:
                    break
    enc.close()
    # the reference writes the file header once per encoder, so the second image of a reused encoder has none: compare its tail
    ref_other, _ = _expected(other, shift_x=1, shift_y=1)
    assert bytes(out) == ref_other[len(ref_other) - len(out):] and len(out) > 1000
    enc = api.Encoder(lib)
    enc.check(enc.set_metadata(1300, 1100, 0, 0, 0))
    enc.check(enc.set_tile_pipeline(8))
    enc.check(enc.provide_output(buf))
    for t in range(5):
        enc.check(enc.send_tile(img, t, 0, 256, 256))
    enc.close()                                           # destroyed with five frames in flight
    assert api.encode_image(lib, img, shift_x=0, shift_y=0) == want


def test_two_encoders_on_two_threads(lib, image):
    """Distinct encoders may run on distinct threads (SURVEY 8b threading contract): the parked
    context, the LF-metadata cache and the staging threads are shared process state."""
    import threading

    imgs = [image("photo", 2048 + 100, 600, 8), image("smooth", 900, 2048 + 64, 16)]
    want = [_expected(i)[0] for i in imgs]
    got = [None, None]
    errs = []

    def work(k):
        try:
            for _ in range(3):
                got[k] = api.encode_image(lib, imgs[k])
        except Exception as e:  # noqa: BLE001 - surfaced below
            errs.append(e)

    ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    assert got[0] == want[0] and got[1] == want[1]


def test_random_shapes_and_modes(lib):
    """Seeded sweep over ragged sizes, sample types, layouts and tile shifts (the host-pointer path has
    per-tile kernel launches, early LF runs every fourth tile and staging helpers: many seams)."""
    from hydrium_amd import synth

    rng = np.random.default_rng(20260928)
    kinds = ["photo", "smooth", "noise", "ramp", "black"]
    for case in range(28):
        w = int(rng.choice([rng.integers(1, 40), rng.integers(200, 700), rng.integers(2040, 2600), rng.integers(4090, 4300)]))
        h = int(rng.choice([rng.integers(1, 40), rng.integers(200, 700), rng.integers(2040, 2200)]))
        depth = int(rng.choice([8, 16, 32]))
        kind = kinds[int(rng.integers(len(kinds)))]
        if depth == 32:
            w, h = min(w, 600), min(h, 600)
            img = synth.make_image_f32(kind, w, h, seed=case)
        else:
            img = synth.make_image(kind, w, h, depth, seed=case)
        kw = {}
        mode = int(rng.integers(4))
        if mode == 1 and max(w, h) <= 2600:
            s = int(rng.integers(0, 4))
            kw = dict(shift_x=s, shift_y=int(rng.integers(0, 4)))
        elif mode == 2:
            kw = dict(layout="planar")
        elif mode == 3 and depth != 32:
            kw = dict(layout="flipped")
        want, _ = _expected(img, **kw)
        got = api.encode_image(lib, img, **kw)
        assert got == want, (case, kind, w, h, depth, kw)


def test_worst_case_content_full_lf_groups(lib):
    """Uniform noise is the format's worst case (2.9 symbols and 14.5 bits per pixel): token buffers,
    bit buffers and the LF coder's records run close to their hard maxima in full 2048-pixel LF groups."""
    from hydrium_amd import synth

    img = synth.make_image("noise", 4096, 4096, 8)
    want, _ = _expected(img)
    got = api.encode_image(lib, img)
    assert len(got) > 28_000_000
    assert got == want


def _encode_documented_protocol(lib, img, out_buf_size, shift=-1, tile_pipeline=None):
    """The call pattern libhydrium.h documents for hyd_send_tile (reference libhydrium.h:222-226):
    flush ONLY while the previous call said HYD_NEED_MORE_OUTPUT.  (The reference's own hyd_send_tile
    never says so — it drops its closing flush's status, libhydrium.c:193-202 — so this client is only
    correct against an implementation that honours the header.)"""
    import ctypes as C

    h, w, _ = img.shape
    tw, th = api.tile_dims(w, h, shift, shift)
    out = bytearray()
    with api.Encoder(lib) as enc:
        enc.check(enc.set_metadata(w, h, 0, shift, shift))
        if tile_pipeline is not None:
            enc.check(enc.set_tile_pipeline(tile_pipeline))
        buf = (C.c_uint8 * out_buf_size)()
        enc.check(enc.provide_output(buf))
        for ty in range(-(-h // th)):
            for tx in range(-(-w // tw)):
                ret = enc.check(enc.send_tile(img, tx, ty, tw, th))
                while ret == api.HYD_NEED_MORE_OUTPUT:
                    code, n = enc.release_output()
                    enc.check(code)
                    out += C.string_at(buf, n)
                    enc.check(enc.provide_output(buf))
                    ret = enc.check(enc.flush())
        code, n = enc.release_output()
        enc.check(code)
        out += C.string_at(buf, n)
    return bytes(out)


@pytest.mark.parametrize("shift,size", [(-1, 64), (-1, 4096), (0, 64), (0, 1 << 20)])
def test_send_tile_reports_pending_output_as_documented(lib, image, shift, size):
    img = image("photo", 600, 520, 8)
    want, _ = _expected(img, shift_x=shift, shift_y=shift)
    assert _encode_documented_protocol(lib, img, size, shift) == want


def test_final_tile_without_an_output_buffer_is_an_api_error(lib, image):
    """reference: hyd_encode_xyb_buffer ends in hyd_flush, whose "buffer was never provided" error
    reaches the caller of hyd_send_tile (encoder.c:1008, libhydrium.c:149-152,193-195)"""
    img = image("photo", 64, 64, 8)
    with api.Encoder(lib) as enc:
        enc.check(enc.set_metadata(64, 64))
        assert enc.send_tile(img, 0, 0, 2048, 2048) == api.HYD_API_ERROR
        assert enc.error_message() == "buffer was never provided"


def test_a_tile_sent_twice_is_rejected(lib, image):
    import ctypes as C

    img = image("photo", 2100, 16, 8)
    with api.Encoder(lib) as enc:
        enc.check(enc.set_metadata(2100, 16))
        buf = (C.c_uint8 * 65536)()
        enc.check(enc.provide_output(buf))
        assert enc.send_tile(img, 0, 0, 2048, 2048, is_last=0) == api.HYD_OK
        assert enc.send_tile(img, 0, 0, 2048, 2048, is_last=1) == api.HYD_API_ERROR
        assert enc.error_message() == "this tile was already sent"
        # the encoder is still usable: the missing tile completes the frame
        enc.check(enc.send_tile(img, 1, 0, 2048, 2048, is_last=1))


_MIXED = r"""
import ctypes as C, hashlib, sys
import numpy as np
from hydrium_amd import api, synth
from oracle import refprobe

def encode(lib, tiles):
    out = bytearray()
    with api.Encoder(lib) as enc:
        enc.check(enc.set_metadata(2048 + 300, 200))
        buf = (C.c_uint8 * (1 << 20))()
        enc.check(enc.provide_output(buf))
        for tx, img in enumerate(tiles):
            h, w, _ = img.shape
            isz = img.dtype.itemsize
            p = img.ctypes.data
            enc.check(enc.send_tile_ptrs([p, p + isz, p + 2 * isz], tx, 0, 3 * w, 3, -1, api._FMT[img.dtype]))
            while True:
                ret = enc.check(enc.flush())
                code, n = enc.release_output()
                out += C.string_at(buf, n)
                enc.check(enc.provide_output(buf))
                if ret != api.HYD_NEED_MORE_OUTPUT:
                    break
    return bytes(out)

a = synth.make_image("photo", 2048, 200, 8)
b = synth.make_image_f32("photo", 300, 200)
got = encode(api.Library(), [a, b])
print(len(got), hashlib.md5(got).hexdigest())
import os
assert refprobe.available() or os.environ.get("HYDAMD_ALLOW_ORACLE_ANCHOR") == "1", "oracle/_ref is absent"
if refprobe.available():
    want = encode(refprobe.reference_library(), [a, b])
    assert got == want, "mixed-format frame differs from the reference"
    print("matches reference")
"""


@pytest.mark.parametrize("eager", ["1", "0"])
def test_sample_format_may_change_between_tiles(eager):
    """a u8 tile followed by an f32 tile grows the staging arena mid-frame; with the
    transforms deferred (HYDAMD_EAGER=0) the first tile's pixels used to be freed before they were read."""
    import subprocess
    import sys

    env = dict(os.environ, HYDAMD_EAGER=eager, PYTHONPATH=os.path.dirname(os.path.dirname(__file__)))
    r = subprocess.run([sys.executable, "-c", _MIXED], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.parametrize("width,height,tiles", [((1 << 20) + 256, 256, [(0, 0), (4096, 0)]), (20000, 20000, [(3, 5), (78, 78)])])
def test_level10_container_through_the_api(lib, image, width, height, tiles):
    """images past level 5 (a side over 2^20, or over 2^28 pixels): container prologue + tile-mode frames"""
    from test_host_glue import _two_tiles_of_a_huge_image
    from oracle import refprobe

    imgs = []
    for k, (tx, ty) in enumerate(tiles):
        tw, th = min(256, width - tx * 256), min(256, height - ty * 256)
        imgs.append(np.ascontiguousarray(image("photo", tw, th, 8, seed=50 + k)))
    got = _two_tiles_of_a_huge_image(lib, width, height, tiles, imgs)
    assert got[4:12] == b"JXL \r\n\x87\n" and b"jxll\x0a" in got[:64]
    if reference_expected():
        assert got == _two_tiles_of_a_huge_image(refprobe.reference_library(), width, height, tiles, imgs)


def test_c5_batch_of_4k_frames_on_four_threads_matches_the_reference(lib):
    """BASELINE configs[4] beyond frame 0: eight different 3840x2160 frames, each encoded twice, on four
    threads sharing the GPU (one encoder per frame, parked device contexts); every file against the reference."""
    import threading

    from hydrium_amd import synth
    from oracle import refprobe

    imgs = [synth.make_image("photo", 3840, 2160, 8, seed=1234 + k) for k in range(8)]
    got = [None] * 16

    def work(t):
        for f in range(t, 16, 4):
            got[f] = hashlib.md5(api.encode_image(lib, imgs[f % 8])).hexdigest()

    ts = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert all(got[f] == got[f % 8] for f in range(16))
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        want0 = [e["md5"] for e in json.load(f)["files"] if (e["kind"], e["width"], e["height"]) == ("photo", 3840, 2160)][0]
    assert got[0] == want0
    if reference_expected():
        ref = refprobe.reference_library(optimised=True)
        for k in range(8):
            assert got[k] == hashlib.md5(api.encode_image(ref, imgs[k])).hexdigest(), f"frame {k}"


_THREADS = r"""
import hashlib, threading
from hydrium_amd import api, synth
from oracle import refprobe
imgs = [synth.make_image("photo", 2300, 2100, 8, seed=s) for s in range(3)] + [synth.make_image("photo", 1000, 4100, 16, seed=9).view("uint16")]
lib = api.Library()
got = {}
def work(t):
    for k in range(6):
        i = (t + k) % len(imgs)
        got[(t, k)] = (i, api.encode_image(lib, imgs[i]))
ts = [threading.Thread(target=work, args=(t,)) for t in range(6)]
[t.start() for t in ts]; [t.join() for t in ts]
import os
assert refprobe.available() or os.environ.get("HYDAMD_ALLOW_ORACLE_ANCHOR") == "1", "oracle/_ref is absent"
ref = refprobe.reference_library(optimised=True) if refprobe.available() else lib
want = [api.encode_image(ref, im) for im in imgs]
assert all(data == want[i] for i, data in got.values())
print("ok", len(got))
"""


@pytest.mark.parametrize("copy_streams", ["0", "1", "4"])
def test_upload_streams_shared_by_the_contexts_of_a_device(copy_streams):
    """round 4: the uploads of all contexts of a device go through HYDAMD_COPY_STREAMS shared streams (0 = a stream per
    context, as before).  Six encoder threads, six frames each of mixed shapes and sample types, every file against
    the reference — one shared stream is the hard case: every context's uploads and fences interleave on it."""
    import subprocess
    import sys

    env = dict(os.environ, HYDAMD_COPY_STREAMS=copy_streams, GPU_MAX_HW_QUEUES="22",
               PYTHONPATH=os.path.dirname(os.path.dirname(__file__)))
    r = subprocess.run([sys.executable, "-c", _THREADS], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0 and "ok 36" in r.stdout, r.stdout + r.stderr


def test_tile_pipeline_request_in_the_middle_of_a_one_frame_image_changes_nothing(lib, image):
    """hydamd_set_tile_pipeline used to hand the encoder's device context back whenever the depth changed — also
    in the middle of a one-frame image, whose uploaded tiles it holds.  The ring belongs to tile mode: a one-frame encoder
    only notes the request."""
    import ctypes as C

    img = image("photo", 4300, 2100, 8)
    want, _ = _expected(img)
    enc = api.Encoder(lib)
    enc.check(enc.set_metadata(4300, 2100, 0, -1, -1))
    buf = (C.c_uint8 * (8 << 20))()
    enc.check(enc.provide_output(buf))
    out = bytearray()
    tiles = [(0, 0), (1, 0), (2, 0), (0, 1), (1, 1), (2, 1)]
    for i, (tx, ty) in enumerate(tiles):
        enc.check(enc.send_tile(img, tx, ty, 2048, 2048))
        if i == 2:
            enc.check(enc.set_tile_pipeline(4))   # three tiles uploaded: their context must survive this
            enc.check(enc.set_tile_pipeline(1))
    while True:
        ret = enc.check(enc.flush())
        code, n = enc.release_output()
        enc.check(code)
        out += C.string_at(buf, n)
        enc.check(enc.provide_output(buf))
        if ret != api.HYD_NEED_MORE_OUTPUT:
            break
    enc.close()
    assert bytes(out) == want


def test_in_place_output_returns_the_callers_buffer(lib, image):
    """api.encode_image(in_place=True) on a multi-tile one-frame image: the file, header included, is found in the caller's
    buffer (a memoryview of it comes back, no copy) — what bench.py's api_end_to_end leg relies on"""
    import ctypes as C

    img = image("photo", 4300, 2100, 8)
    want, _ = _expected(img)
    buf = (C.c_uint8 * (8 << 20))()
    got = api.encode_image(lib, img, out_buf=buf, in_place=True)
    assert isinstance(got, memoryview)
    assert bytes(got) == want


def test_tile_frame_rerun_for_overflow_after_its_results_were_staged():
    """Tile-mode frames leave the device as one staged blob (hydamd_stage_frame_blob / hydamd_read_frame_blob).  A frame that
    outgrows its buffers is rerun inside hydamd_sync with larger ones AFTER it was staged: the staged blob is the first
    run's, and the rerun's results may not even fit the staging buffer (round 5: that was an error until the fuzz sweep found
    it) — the encoder then reads the results the separate way.  Small buffers force the rerun on every noise tile."""
    import subprocess
    import sys

    code = """
import numpy as np
from hydrium_amd import api, synth
from oracle import refprobe
img = synth.make_image("noise", 600, 300, 8)
for kw in (dict(shift_x=0, shift_y=0), dict(shift_x=1, shift_y=1), dict()):
    want = api.encode_image(refprobe.reference_library(), img, **kw)
    for _ in range(2):       # the second pass meets contexts already enlarged
        assert api.encode_image(api.Library(), img, **kw) == want, kw
print("ok")
"""
    assert reference_expected()
    env = dict(os.environ, HYDAMD_TOKEN_CAP="4096", HYDAMD_PAYLOAD_CAP="16384",
               PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
