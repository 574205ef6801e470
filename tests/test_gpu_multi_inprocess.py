"""GPU: one device-resident frame on N devices of ONE process, composed in C — hydamd_multi_create /
hydamd_encode_image_multi / hydamd_multi_result / hydamd_multi_read (csrc/host/multi.c; VERDICT r5 task 6).  It is
hyd_send_tile's multi-device closing stage (reference libhydrium.c:172-203, encoder.c:928-957) without the uploads.
A box with one GPU runs it with the device list aliased (0,0,0,0: four contexts, four streams, every cross-context step
— floor by "peer" read, views, cross-stream waits — only the xGMI hop missing); every file is compared with the
reference's for the same pixels."""
import hashlib

import numpy as np
import pytest

from conftest import has_gpu, reference_expected
from hydrium_amd import api

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs an MI355X")]


def _image(kind, w, h, depth):
    import torch
    from hydrium_amd import synth

    if depth == 32:
        host = synth.make_image_f32(kind, w, h)
        return torch.from_numpy(host).cuda(), host
    t = synth.make_image(kind, w, h, depth, device="cuda")
    torch.cuda.synchronize()
    a = t.cpu().numpy()
    return t, np.ascontiguousarray(a.view(np.uint16) if depth == 16 else a)


def _reference(host):
    from oracle import refprobe

    assert reference_expected()
    return api.encode_image(refprobe.reference_library(optimised=True), host)


@pytest.mark.parametrize("case,devices", [(("photo", 16384, 16384, 8), [0, 0, 0, 0]), (("photo", 8192, 6200, 16), [0, 0, 0]),
                                          (("photo", 4296, 4168, 8), [0, 0]), (("photo", 4100, 4100, 32), [0, 0, 0, 0])],
                         ids=["16384x16384-4shards", "8192x6200-u16-3shards", "4296x4168-2shards", "4100x4100-f32-4shards"])
def test_device_resident_frame_on_an_aliased_device_list_equals_the_reference(case, devices):
    from hydrium_amd import device

    kind, w, h, depth = case
    t, host = _image(kind, w, h, depth)
    want = _reference(host)
    with device.MultiFrame(devices, w, h) as m:
        covered = []
        for d in range(len(devices)):
            first, count = m.shard_lf_groups(d)
            covered += list(range(first, first + count))
        assert covered == list(range((-(-w // 2048)) * (-(-h // 2048))))  # raster runs, every LF group once
        files = []
        for asm in range(len(devices)):  # the assembling shard rotates; the file does not care
            m.encode([t] * len(devices), assembling_shard=asm)
            files.append(bytes(m.read()))
    assert all(f == want for f in files), [hashlib.md5(f).hexdigest() for f in files] + [hashlib.md5(want).hexdigest()]


def test_a_shard_that_reruns_sends_the_later_shards_round_again_and_slabs_need_not_hold_the_whole_image(monkeypatch):
    """(a) float noise with small token arrays: an early shard outgrows them and reruns inside its sync — the alphabet
    maxima it left the first time were incomplete, the later shards read their floor again and run again; (b) every
    shard's pixels in a buffer of its own that holds only its LF groups' rows (the rank-style slab: origin = slab - y0 *
    row stride)."""
    import torch
    from hydrium_amd import device

    monkeypatch.setenv("HYDAMD_TOKEN_CAP", "40000")
    t, host = _image("noise", 4100, 4100, 32)
    want = _reference(host)
    with device.MultiFrame([0, 0, 0], 4100, 4100) as m:
        m.encode([t, t, t], assembling_shard=1)
        got = bytes(m.read())
        assert sum(m.context_overflow_reruns(d) for d in range(3)) >= 1, "the case did not exercise the rerun"
    assert got == want
    monkeypatch.delenv("HYDAMD_TOKEN_CAP")
    t, host = _image("photo", 4296, 6200, 8)  # 3 x 4 LF groups: shards of 6 groups = two LF-group rows each
    want = _reference(host)
    with device.MultiFrame([0, 0], 4296, 6200) as m:
        m.sample_bytes = 1
        slabs, origins = [], []
        for d in range(2):
            first, count = m.shard_lf_groups(d)
            y0, y1 = (first // 3) * 2048, min(6200, ((first + count - 1) // 3 + 1) * 2048)
            slab = t[y0:y1].clone()
            slabs.append(slab)
            origins.append(slab.data_ptr() - y0 * 4296 * 3)
        torch.cuda.synchronize()
        m.encode(origins, assembling_shard=1)
        assert bytes(m.read()) == want


def test_argument_and_protocol_errors():
    from hydrium_amd import device

    with pytest.raises(device.DeviceError):
        device.MultiFrame([0] * 9, 4296, 4168)          # more shards than HYDAMD_MAX_PEERS
    with pytest.raises(device.DeviceError):
        device.MultiFrame([0, 0, 0], 2048, 2048)        # fewer LF groups than shards
    t, _ = _image("photo", 4296, 4168, 8)
    with device.MultiFrame([0, 0], 4296, 4168) as m:
        with pytest.raises(device.DeviceError, match="no frame in flight"):
            m.result()
        m.encode([t, t])
        with pytest.raises(device.DeviceError, match="in flight"):
            m.encode([t, t])
        assert m.result() > 0
        with pytest.raises(device.DeviceError):
            m.encode([t, t], assembling_shard=2)


def test_nan_sample_is_an_api_error_also_on_the_first_frame_whose_peer_reads_are_being_verified():
    import torch
    from hydrium_amd import device, synth

    host = synth.make_image_f32("photo", 4100, 4100).copy()
    host[2050, 2050, 1] = np.nan
    t = torch.from_numpy(host).cuda()
    with device.MultiFrame([0, 0, 0], 4100, 4100) as m:
        m.encode([t, t, t], assembling_shard=0)
        with pytest.raises(device.DeviceError, match="NaN") as ei:
            m.result()
        assert ei.value.code == -14  # HYD_API_ERROR
