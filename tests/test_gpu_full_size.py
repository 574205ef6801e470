"""GPU, BASELINE.json's full-size configurations.

The CPU oracle is too slow for these sizes, so the checks are (a) the real reference when the
prebuilt oracle/_ref travelled (one CPU pass, tens of seconds), and (b) size-independent
properties that hold for any correct encoder of this format:
  * partition invariance — coding the frame as 1, 2, 4 or 8 shards gives the same bytes;
  * path invariance — host-pointer API, device-resident API and every entropy-stage form agree;
  * structural checksum — the TOC's section sizes add up to the file, and the packed payload
    equals the concatenation of the per-group sections the device reports.
"""
import hashlib

import numpy as np
import pytest

from conftest import has_gpu, reference_expected
from hydrium_amd import api

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]


def _device_image(kind, w, h, depth):
    import torch
    from hydrium_amd import synth

    t = synth.make_image(kind, w, h, depth, device="cuda")
    torch.cuda.synchronize()
    return t


def _host(t, depth):
    a = t.cpu().numpy()
    return np.ascontiguousarray(a.view(np.uint16) if depth == 16 else a)


def test_c3_8192_rgb16_all_paths_agree():
    from hydrium_amd import device, multigpu
    from oracle import refprobe

    t = _device_image("photo", 8192, 8192, 16)
    host = _host(t, 16)
    whole = api.encode_image(api.Library(), host)                     # drop-in API, host pixels
    assert multigpu.encode_serial(t, 1) == whole                      # device-resident, one context
    assert multigpu.encode_serial(t, 4) == whole                      # 4 shards
    payloads = []
    with device.DeviceContext(0, 16, 0) as ctx:
        for form in (4, 5, 6):                             # every entropy-stage form
            ctx.set_rans_waves(form)
            ctx.encode_image_tensor(t)
            ctx.sync()
            payloads.append(hashlib.md5(ctx.read_payload()).hexdigest())
            bits = np.concatenate([ctx.read_sections(s)[0] for s in range(16)])
            assert int(((bits + 7) // 8).sum()) == ctx.payload_size()
    assert len(set(payloads)) == 1
    if reference_expected():
        assert whole == api.encode_image(refprobe.reference_library(optimised=True), host)


def test_c4_16384_rgb8_sharded_eight_ways():
    from hydrium_amd import multigpu
    from oracle import refprobe

    t = _device_image("smooth", 16384, 16384, 8)                      # 64 LF groups, 4096 groups
    one = multigpu.encode_serial(t, 1)
    assert one[:2] == b"\xff\x0a"
    eight = multigpu.encode_serial(t, 8)
    assert eight == one
    if reference_expected():
        ref = api.encode_image(refprobe.reference_library(optimised=True), _host(t, 8))
        assert hashlib.md5(one).hexdigest() == hashlib.md5(ref).hexdigest()


def test_c5_4k_frames_batch():
    """Eight independent 3840x2160 frames, round-robin over two contexts; frame 0 against the golden MD5."""
    import json
    import os

    from hydrium_amd import device, multigpu

    with open(os.path.join(os.path.dirname(__file__), "golden", "manifest.json")) as f:
        gold = [e for e in json.load(f)["files"] if (e["width"], e["height"], e["kind"]) == (3840, 2160, "photo")][0]
    t0 = _device_image("photo", 3840, 2160, 8)
    first = multigpu.encode_serial(t0, 1)
    assert (len(first), hashlib.md5(first).hexdigest()) == (gold["size"], gold["md5"])
    frames = [t0] + [_device_image("photo", 3840, 2160, 8).roll(i * 37, 0) for i in range(1, 8)]
    frames = [f.contiguous() for f in frames]
    ctxs = [device.DeviceContext(0, 4, 0) for _ in range(2)]
    try:
        sums = []
        for i in range(0, 8, 2):
            for j in range(2):
                ctxs[j].encode_image_tensor(frames[i + j])
            for j in range(2):
                ctxs[j].sync()
                sums.append(hashlib.md5(ctxs[j].read_payload()).hexdigest())
        again = []
        for f in frames:                                               # one at a time on one context
            ctxs[0].encode_image_tensor(f)
            ctxs[0].sync()
            again.append(hashlib.md5(ctxs[0].read_payload()).hexdigest())
        assert sums == again
    finally:
        for c in ctxs:
            c.close()
