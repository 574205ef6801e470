"""GPU: the device-side frame assembler (csrc/hip/assemble.hip, hydamd_assembler_*) must produce, from
shard blobs left in device memory, the very bytes the host assembler makes of the same blobs
(hydamd_frame_from_blobs) — and those are the reference's (whole files through hyd_send_tile, and the
compiled reference itself when oracle/_ref travelled)."""
import hashlib

import numpy as np
import pytest

from conftest import has_gpu, reference_expected
from hydrium_amd import api

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]


def _cuda(img):
    import torch

    if img.dtype == np.uint16:
        return torch.from_numpy(img.view(np.int16).copy()).cuda()
    return torch.from_numpy(np.ascontiguousarray(img)).cuda()


def _blobs_on_device(t, w, h, parts, linear_light=0):
    """Code the frame as len(parts) shards on one GPU, floors in shard order; returns the blobs as CUDA tensors."""
    import torch
    from hydrium_amd import device, multigpu

    lfx = -(-w // 2048)
    engines = [multigpu.GpuShardEngine(multigpu.Shard(0, p, w, h, linear_light), t, lambda lf: ((lf // lfx) * 2048, (lf % lfx) * 2048))
               for p in parts]
    blobs = []
    try:
        for e in engines:
            e.enqueue_transform()
        torch.cuda.synchronize()
        maxima = [[int(v) for v in e.alphabet_maxima().cpu()] for e in engines]
        seen = []
        for e, mx in zip(engines, maxima):
            floor = torch.tensor([max(seen, default=0)], dtype=torch.int32, device="cuda")
            seen += mx
            e.enqueue_entropy(floor)
            for attempt in range(2):  # a frame that outgrows the context's buffers is rerun inside finish(): export again
                out = torch.zeros(e.blob_bound(), dtype=torch.uint8, device="cuda")
                e.export_blob(out)
                e.finish()
                head = device.blob_header(out[:64].cpu().numpy().tobytes())
                if not int(head["status"]) & device.BLOB_RETRY:
                    break
            assert int(head["status"]) == 0
            blobs.append(out[: (int(head["total_bytes"]) + 15) & ~15].clone())
    finally:
        for e in engines:
            e.shard.close()
    torch.cuda.synchronize()
    return blobs


def _assemble_on_device(md, blobs, parts, out_cap=None, pinned=False, **kw):
    import torch
    from hydrium_amd import device

    cap = out_cap or sum(b.numel() for b in blobs) + (1 << 20)
    out = torch.zeros(cap, dtype=torch.uint8).pin_memory() if pinned else torch.full((cap,), 0xA5, dtype=torch.uint8, device="cuda")
    with device.Assembler(0) as asm:
        asm.plan(md, parts, **kw)
        asm.run_tensors(blobs, out)
        torch.cuda.synchronize()
        n = asm.result()
        data = bytes(out[:n].cpu().numpy()) if not pinned else bytes(out[:n].numpy())
        if not pinned:
            assert bool((out[n:n + 64].cpu() == 0xA5).all()), "the assembler wrote past the frame's end"
        return data


def _host_assembly(md, blobs, **kw):
    from hydrium_amd import device

    return device.frame_from_blobs(md, [b.cpu().numpy().tobytes() for b in blobs], **kw)


CASES = [
    # kind, w, h, depth, shards
    ("photo", 4096 + 200, 2 * 2048 + 72, 8, 1),   # 3 x 3 LF groups, ragged right and bottom: four LF group shapes
    ("photo", 4096 + 200, 2 * 2048 + 72, 8, 4),
    ("photo", 2048 + 300, 2048 + 40, 16, 2),
    ("photo", 700, 500, 8, 1),                    # one LF group, six groups
    ("smooth", 2048 + 8, 16, 8, 2),               # two LF groups of one block row: tiny LF streams (simple prefix codes)
    ("noise", 1000, 600, 8, 1),                   # large sections
    ("smooth", 4096, 4096, 8, 3),
]


@pytest.mark.parametrize("kind,w,h,depth,shards", CASES)
def test_device_assembly_equals_host_assembly_and_the_api(image, kind, w, h, depth, shards):
    from hydrium_amd import sharding
    from oracle import refprobe

    img = image(kind, w, h, depth)
    n_lf = (-(-w // 2048)) * (-(-h // 2048))
    parts = [p for p in sharding.partition_lf_groups(n_lf, shards) if p]
    blobs = _blobs_on_device(_cuda(img), w, h, parts)
    md = api.HYDImageMetadata(w, h, 0, -1, -1)
    got = _assemble_on_device(md, blobs, parts)
    assert got == _host_assembly(md, blobs)
    assert got == api.encode_image(api.Library(), img)
    if reference_expected() and w * h <= 2400 * 2400:
        assert got == api.encode_image(refprobe.reference_library(), img)


def test_frames_without_file_header_and_not_last(image):
    w, h = 2048 + 100, 300
    img = image("photo", w, h, 8)
    parts = [[0], [1]]
    blobs = _blobs_on_device(_cuda(img), w, h, parts)
    md = api.HYDImageMetadata(w, h, 0, -1, -1)
    for kw in (dict(write_header=False), dict(is_last=False), dict(write_header=False, is_last=False)):
        assert _assemble_on_device(md, blobs, parts, **kw) == _host_assembly(md, blobs, **kw)


def test_icc_profile_and_pinned_host_output(image):
    w, h = 1200, 520
    img = image("photo", w, h, 8)
    icc = bytes(range(256)) * 3 + b"tail"
    blobs = _blobs_on_device(_cuda(img), w, h, [[0]])
    md = api.HYDImageMetadata(w, h, 0, -1, -1)
    got = _assemble_on_device(md, blobs, [[0]], icc=icc, pinned=True)
    assert got == _host_assembly(md, blobs, icc=icc)
    assert got == api.encode_image(api.Library(), img, icc=icc)


def test_send_order_other_than_raster(image):
    """Blobs in a different order: the frame's section order and TOC permutation follow the blobs
    (reference encoder.c:241-325), here checked against the host assembler fed the same way."""
    w, h = 4096 + 64, 2048 + 64
    img = image("photo", w, h, 8)
    parts = [[4, 5], [0, 1], [2, 3]]  # shard k codes these LF groups; floors follow this order too
    blobs = _blobs_on_device(_cuda(img), w, h, parts)
    md = api.HYDImageMetadata(w, h, 0, -1, -1)
    got = _assemble_on_device(md, blobs, parts)
    assert got == _host_assembly(md, blobs)
    order = [(lf % 3, lf // 3) for p in parts for lf in p]
    assert got == api.encode_image(api.Library(), img, order=order)


def test_float_frame_with_large_alphabet():
    import torch
    from hydrium_amd import synth

    w, h = 2048 + 64, 600
    img = synth.make_image_f32("photo", w, h)
    img[:64, :64] *= 300.0  # out-of-gamut: tokens above 32, log alphabet size above 5
    t = torch.from_numpy(img).cuda()
    blobs = _blobs_on_device(t, w, h, [[0], [1]])
    md = api.HYDImageMetadata(w, h, 0, -1, -1)
    got = _assemble_on_device(md, blobs, [[0], [1]])
    assert got == _host_assembly(md, blobs)
    assert got == api.encode_image(api.Library(), img)


def test_assembler_reports_what_it_cannot_do(image):
    import torch
    from hydrium_amd import device

    w, h = 2048 + 100, 300
    img = image("photo", w, h, 8)
    parts = [[0], [1]]
    blobs = _blobs_on_device(_cuda(img), w, h, parts)
    md = api.HYDImageMetadata(w, h, 0, -1, -1)
    want = _host_assembly(md, blobs)
    with device.Assembler(0) as asm:
        with pytest.raises(device.DeviceError):  # a single-group frame stays with the host assembler
            asm.plan(api.HYDImageMetadata(200, 100, 0, -1, -1), [[0]])
        with pytest.raises(device.DeviceError):  # an LF group twice
            asm.plan(md, [[0], [0]])
        asm.plan(md, parts)
        small = torch.zeros(1024, dtype=torch.uint8, device="cuda")
        asm.run_tensors(blobs, small)
        torch.cuda.synchronize()
        with pytest.raises(device.DeviceError) as ei:  # output too small: says how much it needs
            asm.result()
        assert ei.value.code == -2
        asm.run_tensors(blobs[::-1], torch.zeros(len(want) + 64, dtype=torch.uint8, device="cuda"))  # blobs swapped: presets disagree
        torch.cuda.synchronize()
        with pytest.raises(device.DeviceError):
            asm.result()
        broken = [b.clone() for b in blobs]
        broken[1][12:16] = torch.tensor([2, 0, 0, 0], dtype=torch.uint8)  # status: the frame outgrew a buffer
        asm.run_tensors(broken, torch.zeros(len(want) + 64, dtype=torch.uint8, device="cuda"))
        torch.cuda.synchronize()
        with pytest.raises(device.DeviceError):
            asm.result()
        out = torch.zeros(len(want), dtype=torch.uint8, device="cuda")  # and afterwards it still works, into an exact-size buffer
        asm.run_tensors(blobs, out)
        torch.cuda.synchronize()
        assert asm.result() == len(want) and bytes(out.cpu().numpy()) == want


def test_c4_16384_photo_assembled_on_the_device_equals_the_reference():
    """BASELINE configs[3] on the bench's own content: 64 LF groups as eight shards' blobs, assembled on the
    device, against the compiled reference (one -O2 CPU pass) or, without it, the host assembler."""
    import torch
    from hydrium_amd import sharding, synth
    from oracle import refprobe

    w = h = 16384
    t = synth.make_image("photo", w, h, 8, device="cuda")
    torch.cuda.synchronize()
    parts = sharding.partition_lf_groups(64, 8)
    blobs = _blobs_on_device(t, w, h, parts)
    md = api.HYDImageMetadata(w, h, 0, -1, -1)
    got = _assemble_on_device(md, blobs, parts)
    assert hashlib.md5(got).hexdigest() == hashlib.md5(_host_assembly(md, blobs)).hexdigest()
    if reference_expected():
        ref = api.encode_image(refprobe.reference_library(optimised=True), np.ascontiguousarray(t.cpu().numpy()))
        assert (len(got), hashlib.md5(got).hexdigest()) == (len(ref), hashlib.md5(ref).hexdigest())


def test_a_context_s_own_view_blob(image):
    """hydamd_export_frame_owned: header and slot records only, the sections stay in the context's buffers and the header
    names their addresses — what hyd_send_tile assembles from.  Same file as from the self-contained blob; the host
    assembler refuses a view."""
    import torch
    from hydrium_amd import device

    w, h = 2048 + 333, 2048 + 90
    img = image("photo", w, h, 16)
    t = _cuda(img)
    md = api.HYDImageMetadata(w, h, 0, -1, -1)
    with device.DeviceContext(0, 4, 0) as ctx, device.Assembler(0) as asm:
        ctx.encode_image_tensor(t)
        full = torch.zeros(ctx.blob_bound(4), dtype=torch.uint8, device="cuda")
        ctx.export_frame(4, full)
        ptr, cap = ctx.export_frame_owned(4)
        assert cap < 64 * 1024  # records only
        out = torch.full((ctx.blob_bound(4),), 0x5A, dtype=torch.uint8, device="cuda")
        asm.plan(md, [[0, 1, 2, 3]])
        asm.run([ptr], [cap], out.data_ptr(), out.numel(), ctx.get_stream())
        ctx.sync()
        n = asm.result()
        got = bytes(out[:n].cpu().numpy())
        head = device.blob_header(full[:64].cpu().numpy().tobytes())
        blob = full[: int(head["total_bytes"])].cpu().numpy().tobytes()
    assert got == device.frame_from_blobs(md, [blob])
    assert got == api.encode_image(api.Library(), img)
