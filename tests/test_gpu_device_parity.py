"""GPU parity: the HIP hot path (through the hydamd_* C-ABI) against the CPU oracle, stage by stage.

Bar: bit-exact for everything integer (LF ints, quantised coefficients, tokens, frequencies,
section bytes); the XYB and DCT float intermediates are also required to be bit-identical
(tolerance 0 ulp; +0.0 and -0.0 compare equal, see kernels.hip dct8).
"""
import os

import numpy as np
import pytest

from conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]

CASES = [
    ("photo", 256, 256, 8),
    ("photo", 264, 200, 16),
    ("noise", 64, 48, 8),
    ("smooth", 517, 259, 8),
    ("photo", 97, 301, 16),
    ("ramp", 8, 8, 8),
    ("black", 40, 24, 8),
    ("white", 33, 9, 16),
    ("noise", 256, 256, 16),
    ("photo", 1000, 700, 8),
]


def _torch_image(img):
    import torch

    if img.dtype == np.uint16:
        return torch.from_numpy(img.view(np.int16).copy()).cuda()
    return torch.from_numpy(np.ascontiguousarray(img)).cuda()


def _check_lf_group(ctx, slot, res, scheme_local_offset, check_planes, pitch_rows=None):
    from hydrium_amd import device

    counts = ctx.read_symbol_counts(slot)
    assert np.array_equal(counts[:res.num_groups], res.group_symbols)
    first = 0
    for g in range(res.num_groups):
        n = int(res.group_symbols[g])
        tok, cl, rb, resid = device.decode_token_records(ctx.read_tokens(slot, g, n))
        ref = res.symbols[first:first + n]
        assert np.array_equal(tok, ref["token"]), f"group {g} tokens"
        assert np.array_equal(cl + scheme_local_offset, ref["cluster"]), f"group {g} clusters"
        assert np.array_equal(rb, ref["residue_bits"]), f"group {g} residue bits"
        assert np.array_equal(resid, ref["residue"]), f"group {g} residues"
        first += n
    freq, alpha, log_alpha, run_max = ctx.read_tables(slot)
    ncl = res.cluster_to - res.cluster_from
    assert np.array_equal(alpha[:ncl], res.alphabet_size[res.cluster_from:res.cluster_to])
    assert np.array_equal(freq[:ncl], res.freqs[res.cluster_from:res.cluster_to])
    assert (log_alpha, run_max) == (res.log_alphabet_size, res.max_alphabet_size)
    assert np.array_equal(ctx.read_dc(slot, res.vbw, res.vbh), res.dc)


@pytest.mark.parametrize("xyb_mode", [0, 1, 2])  # registers + v_rcp, registers + IEEE division, LUT gathers
@pytest.mark.parametrize("kind,w,h,depth", CASES)
def test_lf_group_stages_match_oracle(image, kind, w, h, depth, xyb_mode):
    from hydrium_amd import device
    from oracle import binding as orc

    img = image(kind, w, h, depth)
    res, _ = orc.encode_lf_group(img)
    timg = _torch_image(img)
    with device.DeviceContext(0, 1, 0, debug_planes=True) as ctx:
        assert ctx.xyb_mode() == 0, "the fast register evaluation of the format.c LUTs failed its bit-exactness self-test"
        ctx.set_xyb_mode(xyb_mode)
        ctx.encode_image_tensor(timg)
        ctx.sync()
        rows, pitch = res.vbh * 8, res.stride
        xyb = ctx.read_debug_plane(0, pitch, rows)
        assert np.array_equal(xyb.view(np.uint32), res.xyb.view(np.uint32)), "XYB planes differ"
        dct = ctx.read_debug_plane(1, pitch, rows)
        assert np.array_equal(dct, res.dct), "DCT planes differ"
        quant = ctx.read_debug_plane(2, pitch, rows)
        assert np.array_equal(quant, res.quant), "quantised planes differ"
        _check_lf_group(ctx, 0, res, res.cluster_from, True)
        bits, offs = ctx.read_sections(0)
        assert np.array_equal(bits[:res.num_groups], res.group_bits)
        payload = ctx.read_payload()
        assert len(payload) == len(res.stream)
        assert payload == res.stream
        assert np.array_equal(offs[:res.num_groups], res.group_offset)


def test_f32_input_and_nan_rejection(image):
    from hydrium_amd import device, synth
    from oracle import binding as orc
    import torch

    img = synth.make_image_f32("photo", 300, 270)
    res, _ = orc.encode_lf_group(img)
    with device.DeviceContext(0, 1, 0) as ctx:
        t_img = torch.from_numpy(img).cuda()  # must outlive the queued work
        ctx.encode_image_tensor(t_img)
        ctx.sync()
        assert ctx.read_payload() == res.stream
        bad = img.copy()
        bad[7, 9, 2] = np.inf
        t_bad = torch.from_numpy(bad).cuda()
        ctx.encode_image_tensor(t_bad)
        with pytest.raises(device.DeviceError) as ei:
            ctx.sync()
        assert ei.value.code == -14 and "NaN" in ei.value.message


def test_multi_lf_group_frame_host_and_device_paths(image):
    """2x2 LF groups (ragged right/bottom): presets, running alphabet, host-staged == device-resident."""
    from hydrium_amd import device
    from oracle import binding as orc

    img = image("photo", 2048 + 300, 2048 + 120, 8)
    expected = []
    mx = 0
    for ty in range(2):
        for tx in range(2):
            res, mx = orc.encode_lf_group(img, tx, ty, max_alphabet_size=mx)
            expected.append(res)
    want = b"".join(r.stream for r in expected)
    with device.DeviceContext(0, 4, 0) as ctx:
        t_img = _torch_image(img)  # must outlive the queued work
        ctx.encode_image_tensor(t_img)
        ctx.sync()
        got_dev = ctx.read_payload()
        for slot, res in enumerate(expected):
            _check_lf_group(ctx, slot, res, res.cluster_from, False)
        ctx.encode_image_host(img)
        ctx.sync()
        got_host = ctx.read_payload()
    assert got_dev == want
    assert got_host == want


def test_planar_and_strided_inputs(image):
    """Generic (non-packed) addressing: planar channels and RGBA-style pixel stride 4."""
    import torch
    from hydrium_amd import device
    from oracle import binding as orc

    img = image("photo", 200, 136, 8)
    res, _ = orc.encode_lf_group(img)
    h, w, _ = img.shape
    with device.DeviceContext(0, 1, 0) as ctx:
        planes = torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1))).cuda()
        ctx.begin_frame(1)
        p = planes.data_ptr()
        ctx.encode_lf_group(0, [p, p + w * h, p + 2 * w * h], w, 1, 0, w, h, 0)
        ctx.finish_frame(1)
        ctx.sync()
        assert ctx.read_payload() == res.stream
        rgba = np.zeros((h, w, 4), np.uint8)
        rgba[:, :, :3] = img
        t = torch.from_numpy(rgba).cuda()
        ctx.begin_frame(1)
        p = t.data_ptr()
        ctx.encode_lf_group(0, [p, p + 1, p + 2], 4 * w, 4, 0, w, h, 0)
        ctx.finish_frame(1)
        ctx.sync()
        assert ctx.read_payload() == res.stream


@pytest.mark.parametrize("form", [4, 5, 6])
def test_frame_that_outgrows_its_buffers_is_rerun_transparently(form):
    """Token arrays and the payload are sized for typical content; a noise frame (2.9 symbols and 1.8
    bytes per pixel) overflows both on the device and hydamd_sync() reruns it with the hard maxima."""
    import subprocess
    import sys

    code = f"""
import numpy as np, torch
from hydrium_amd import device, synth
from oracle import binding as orc
img = synth.make_image("noise", 520, 300, 8)
want, _ = orc.encode_lf_group(img)
with device.DeviceContext(0, 1, 0) as ctx:
    ctx.set_rans_waves({form})
    assert ctx.token_capacity() == 4096
    t = torch.from_numpy(img).cuda()
    ctx.encode_image_tensor(t); ctx.sync()
    assert ctx.overflow_reruns() >= 1 and ctx.token_capacity() == 196608, (ctx.overflow_reruns(), ctx.token_capacity())
    assert ctx.read_payload() == want.stream
    ctx.encode_image_tensor(t); ctx.sync()            # the context stays enlarged: no second rerun
    assert ctx.overflow_reruns() == 1 or ctx.overflow_reruns() == 2
    n = ctx.overflow_reruns()
    ctx.encode_image_tensor(t); ctx.sync()
    assert ctx.overflow_reruns() == n and ctx.read_payload() == want.stream
print("ok")
"""
    env = dict(os.environ, HYDAMD_TOKEN_CAP="4096", HYDAMD_PAYLOAD_CAP="65536",
               PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.parametrize("form", [4, 5, 6])
@pytest.mark.parametrize("num_presets,scheme", [(28, 0), (29, 1), (85, 1), (86, 2), (127, 2), (129, 3), (255, 3)])
def test_every_cluster_scheme_on_the_device(image, num_presets, scheme, form):
    """the 2- and 1-cluster-per-preset maps (frames of 86 to 255 LF groups, reference
    encoder.c:880-901) had only run on the CPU oracle.  A frame header of `num_presets` presets with three
    small LF groups sent under scattered preset ids: tokens, clusters, tables and section bytes of each
    against the oracle coding the same LF group with the same (preset, num_presets), running alphabet in
    send order."""
    from hydrium_amd import device
    from oracle import binding as orc

    imgs = [image("photo", 300, 200, 8, seed=7), image("noise", 72, 40, 8, seed=8), image("smooth", 520, 264, 16, seed=9)]
    presets = [0, num_presets // 2, num_presets - 1]
    per_preset = {0: 9, 1: 3, 2: 2, 3: 1}[scheme]
    with device.DeviceContext(0, 3, 0) as ctx:
        ctx.set_rans_waves(form)
        ctx.begin_frame(num_presets)
        keep = []
        for slot, (img, p) in enumerate(zip(imgs, presets)):
            t = _torch_image(img)
            keep.append(t)
            isz = t.element_size()
            h, w, _ = img.shape
            base = t.data_ptr()
            ctx.encode_lf_group(slot, [base, base + isz, base + 2 * isz], 3 * w, 3, {1: 0, 2: 1}[isz], w, h, p)
        ctx.finish_frame(3)
        ctx.sync()
        payload = ctx.read_payload()
        running = 0
        at = 0
        for slot, (img, p) in enumerate(zip(imgs, presets)):
            res, running = orc.encode_lf_group(np.ascontiguousarray(img), num_presets=num_presets, preset=p,
                                               max_alphabet_size=running)
            assert res.cluster_to - res.cluster_from == per_preset
            _check_lf_group(ctx, slot, res, res.cluster_from, False)
            bits, offs = ctx.read_sections(slot)
            assert np.array_equal(bits[:res.num_groups], res.group_bits)
            assert int(offs[0]) == at
            assert payload[at:at + len(res.stream)] == res.stream, f"sections of slot {slot}"
            at += len(res.stream)
        assert at == len(payload)


def test_transform_kernel_keeps_its_place_beside_the_entropy_stage():
    """A performance invariant with a cliff behind it: LDS comes in granules of 1280 bytes, a CU has 128 of them and a
    lane-form entropy workgroup takes 75 — two transform workgroups fit beside it only at <= 26 granules each (a build
    52 bytes over lost 4 % of the pipelined rate), and four of them need <= 128 registers per thread."""
    from hydrium_amd import device

    with device.DeviceContext(0, 1, 0) as ctx:
        for fmt in (0, 1, 2):  # u8, u16, f32
            lds, regs = ctx.transform_footprint(fmt)
            assert 0 < lds <= 26 * 1280, (fmt, lds)
            assert 0 < regs <= 128, (fmt, regs)


def test_batch_of_frames_as_one_launch_group(image):
    """hydamd_encode_image_batch: three different pictures of one shape (two LF groups each, ragged) coded as ONE launch
    group — frame k in slots 2k, 2k + 1, presets and the running alphabet maximum restarting with every frame — leave
    exactly the tables, sections and coded LF streams each picture leaves when a context codes it alone (which the other
    tests of this module tie to the oracle and the API tests to the reference)."""
    from hydrium_amd import device

    pics = [_torch_image(image(kind, 2300, 1500, depth, seed=700 + k)) for k, (kind, depth) in
            enumerate([("photo", 8), ("noise", 8), ("smooth", 8)])]
    alone = []
    with device.DeviceContext(0, 2, 0) as ctx:
        ctx.set_rans_waves(5)
        ctx.set_lf_coder(2)
        for t in pics:
            ctx.encode_image_tensor(t)
            ctx.sync()
            lf = ctx.read_lf_streams(2)
            lfp = ctx.read_lf_payload()
            alone.append(dict(payload=ctx.read_payload(), tables=[ctx.read_tables(s) for s in range(2)],
                              sections=[ctx.read_sections(s)[0] for s in range(2)],
                              lf=[(int(r["bit_count"]), bytes(r["lengths"]), bytes(lfp[int(r["offset"]):int(r["offset"]) + (int(r["bit_count"]) + 7) // 8])) for r in lf]))
    with device.DeviceContext(0, 6, 0) as ctx:
        ctx.set_rans_waves(5)
        ctx.set_lf_coder(2)
        for rep in range(2):  # twice: the context's buffers are reused from batch to batch
            assert ctx.encode_image_batch(pics) == 6
            ctx.sync()
            assert ctx.read_payload() == b"".join(a["payload"] for a in alone)
            lf = ctx.read_lf_streams(6)
            lfp = ctx.read_lf_payload()
            for k, a in enumerate(alone):
                for s in range(2):
                    f, al, la, rm = ctx.read_tables(2 * k + s)
                    f0, al0, la0, rm0 = a["tables"][s]
                    assert np.array_equal(f, f0) and np.array_equal(al, al0) and (la, rm) == (la0, rm0), (k, s)
                    assert np.array_equal(ctx.read_sections(2 * k + s)[0], a["sections"][s])
                    r = lf[2 * k + s]
                    got = (int(r["bit_count"]), bytes(r["lengths"]), bytes(lfp[int(r["offset"]):int(r["offset"]) + (int(r["bit_count"]) + 7) // 8]))
                    assert got == a["lf"][s], (k, s)
        with pytest.raises(device.DeviceError):
            ctx.export_frame_owned(6)  # a batch is not one frame's blob
        ctx.encode_image_tensor(pics[0])  # and the context codes single frames again afterwards
        ctx.sync()
        assert ctx.read_payload() == alone[0]["payload"]


@pytest.mark.parametrize("kind,depth", [("noise", 8), ("noise", 16), ("photo", 16)])
def test_curve_gather_choice_never_changes_a_byte(image, kind, depth):
    """The transform kernel reads one of a pixel's six curves from the uploaded table or evaluates all six in registers
    (hydamd_set_curve_gathers; by default chosen from the density of the context's last frame: a noise frame switches the
    NEXT one to registers).  Every choice leaves the oracle's sections."""
    from hydrium_amd import device
    from oracle import binding as orc

    img = image(kind, 520, 300, depth)
    want, _ = orc.encode_lf_group(img)
    t = _torch_image(img)
    with device.DeviceContext(0, 1, 0) as ctx:
        for mode in (1, 2, 0, 0, 0):          # forced both ways, then by content: first frame, and frames that follow one
            ctx.set_curve_gathers(mode)
            ctx.encode_image_tensor(t)
            ctx.sync()
            assert ctx.read_payload() == want.stream, (kind, depth, mode)


@pytest.mark.parametrize("form", [4, 5])
def test_integer_frame_in_a_context_an_earlier_float_frame_widened(image, form):
    """A float LF group re-lays the context's token arrays for 8-byte records and the context stays wide; an integer frame
    that follows still writes 4-byte records.  Small launches split every group over four transform workgroups whose parts
    k_join_parts closes up: the record width there is the LF group's, not the array's (round 5: the first version used the
    array's, and k_rans_emit walked off the end of a section)."""
    from hydrium_amd import device, synth
    from oracle import binding as orc
    import torch

    f = synth.make_image_f32("photo", 264, 136)
    want_f, _ = orc.encode_lf_group(f)
    with device.DeviceContext(0, 1, 0) as ctx:
        ctx.set_rans_waves(form)
        t_f = torch.from_numpy(f).cuda()
        ctx.encode_image_tensor(t_f)
        ctx.sync()
        assert ctx.read_payload() == want_f.stream
        for kind, w, h, depth in (("photo", 72, 40, 8), ("photo", 520, 300, 16), ("noise", 257, 255, 8)):
            img = image(kind, w, h, depth)
            want, _ = orc.encode_lf_group(img)
            t = _torch_image(img)
            ctx.encode_image_tensor(t)
            ctx.sync()
            assert ctx.read_payload() == want.stream, (kind, w, h, depth)
        ctx.encode_image_tensor(t_f)      # and a float frame again
        ctx.sync()
        assert ctx.read_payload() == want_f.stream


def test_contexts_take_the_hint_when_another_one_outgrew_its_buffers():
    """round 6: once a context of the process has rerun a frame because the default arrays were too small (noise: 2.9
    symbols per pixel), an IDLE context enlarges its arrays ahead of its next frame (hydamd_begin_frame) instead of
    finding out by itself; the bytes are the same either way.  A process of its own: the hint is process-wide."""
    import subprocess
    import sys

    code = """
import torch
from hydrium_amd import device, synth
img = synth.make_image("noise", 2048, 2048, 8, device="cuda")
with device.DeviceContext(0, 1, 0) as a, device.DeviceContext(0, 1, 0) as b:
    a.encode_image_tensor(img); a.sync()
    assert a.overflow_reruns() >= 1 and a.grown_ahead() == 0, (a.overflow_reruns(), a.grown_ahead())
    want = a.read_payload()
    b.encode_image_tensor(img); b.sync()
    assert b.overflow_reruns() == 0 and b.grown_ahead() == 1, (b.overflow_reruns(), b.grown_ahead())
    assert b.token_capacity() == a.token_capacity() == 196608
    assert b.read_payload() == want
print("ok")
"""
    env = dict(os.environ, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env.pop("HYDAMD_TOKEN_CAP", None)
    env.pop("HYDAMD_PAYLOAD_CAP", None)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def test_splitting_every_transform_launch_never_changes_a_byte():
    """HYDAMD_K1_SPLIT_SLOTS=n splits the transform launches of up to n LF groups over four (HYDAMD_K1_SPLIT_LOG=1: two)
    workgroups per group, as one- and two-LF-group launches always are (k_join_parts closes the token arrays up): an A/B knob
    of the pipelined loop (profiles/r06_split_loop.txt).  A frame of four LF groups, ragged edges, both sample depths: sections
    and coded LF streams equal to the unsplit launch's, which equals the oracle's (every other test of this file)."""
    import subprocess
    import sys

    code = """
import hashlib, sys, torch
from hydrium_amd import device, synth
out = []
for depth in (8, 16):
    img = synth.make_image("photo", 2300, 2100, depth)
    t = torch.from_numpy(img.view("int16").copy() if depth == 16 else img).cuda()
    with device.DeviceContext(0, 4, 0) as ctx:
        ctx.set_rans_waves(5)
        ctx.encode_image_tensor(t); ctx.sync()
        out.append(hashlib.md5(ctx.read_payload()).hexdigest())
        assert ctx.overflow_reruns() == 0
print("MD5 " + " ".join(out))
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    for name, extra in (("plain", {"HYDAMD_K1_SPLIT_SLOTS": "0"}), ("four parts", {"HYDAMD_K1_SPLIT_SLOTS": "32", "HYDAMD_K1_SPLIT_LOG": "2"}),
                        ("two parts", {"HYDAMD_K1_SPLIT_SLOTS": "32", "HYDAMD_K1_SPLIT_LOG": "1"})):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, PYTHONPATH=root, **extra), timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        got[name] = next(l for l in r.stdout.splitlines() if l.startswith("MD5 "))
    assert got["plain"] == got["four parts"] == got["two parts"], got
