"""GPU: the multi-GPU sharding choreography run on one GPU (N shards, one after another) must give
the same file as the unsharded encode / the reference, and a batch of independent
frames round-robined over contexts (config C5) must give the same files as one at a time."""
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


@pytest.mark.parametrize("shards", [1, 2, 3, 4, 8])
def test_sharded_frame_equals_reference(image, shards):
    from hydrium_amd import multigpu
    from oracle import refprobe

    img = image("photo", 4096 + 200, 2 * 2048 + 72, 8)  # 3 x 3 LF groups, ragged right and bottom
    got = multigpu.encode_serial(_cuda(img), shards)
    whole = api.encode_image(api.Library(), img)
    assert got == whole
    if reference_expected():
        assert got == api.encode_image(refprobe.reference_library(), img)


def test_sharded_frame_with_host_lf_coder(image, monkeypatch):
    """LF coder off: shards ship LF ints and rank 0's host codes them (hydamd_frame_from_results);
    on (the default, all other tests): shards ship coded LF streams (hydamd_frame_from_streams)."""
    from hydrium_amd import multigpu

    img = image("photo", 2048 + 300, 2048 + 40, 16)
    want = multigpu.encode_serial(_cuda(img), 2)
    monkeypatch.setenv("HYDAMD_LF_CODER", "0")
    assert multigpu.encode_serial(_cuda(img), 3) == want


def test_sharded_float_frame_with_growing_alphabet():
    """Out-of-gamut floats make later LF groups need a larger alphabet than earlier ones: the
    running-maximum floor that shards exchange must reproduce the single-context result."""
    import torch
    from hydrium_amd import multigpu, synth

    img = synth.make_image_f32("photo", 2048 + 64, 2048 + 64)
    img[2048:, 2048:] *= 4000.0        # huge coefficients only in the last LF group
    img[:64, :64] *= 300.0
    t = torch.from_numpy(img).cuda()
    one = multigpu.encode_serial(t, 1)
    for shards in (2, 4):
        assert multigpu.encode_serial(t, shards) == one
    assert one == api.encode_image(api.Library(), img)


def test_batch_of_frames_round_robin_contexts(image):
    """Config C5 in miniature: independent frames, frame i on context i mod 2, all queued before any sync."""
    from hydrium_amd import device
    from oracle import binding as orc

    frames = [image("photo", 960, 540, 8, seed=100 + i) for i in range(5)]
    ctxs = [device.DeviceContext(0, 1, 0) for _ in range(2)]
    try:
        got = []
        for i in range(0, len(frames), 2):
            batch = frames[i:i + 2]
            keep = [_cuda(f) for f in batch]  # device pixels must stay alive until the queued work has run
            for j, t in enumerate(keep):
                ctxs[j].encode_image_tensor(t)
            for j, f in enumerate(batch):
                ctxs[j].sync()
                got.append(ctxs[j].read_payload())
    finally:
        for c in ctxs:
            c.close()
    for f, g in zip(frames, got):
        res, _ = orc.encode_lf_group(np.ascontiguousarray(f))
        assert g == res.stream


def test_payload_tensor_and_rccl_gather_world_of_one(image):
    """The pieces bench.py --gpus N strings together, as far as one GPU can exercise them: the
    zero-copy torch view of the packed sections and sharding.all_gather_sections over the "nccl"
    (= RCCL) backend."""
    import socket

    import torch
    import torch.distributed as dist
    from hydrium_amd import device, sharding

    img = image("photo", 1000, 700, 8)
    t = _cuda(img)
    with device.DeviceContext(0, 1, 0) as ctx:
        ctx.encode_image_tensor(t)
        ctx.sync()
        want = ctx.read_payload()
        view = ctx.payload_tensor()
        assert view.is_cuda and view.dtype == torch.uint8 and view.numel() == len(want)
        assert bytes(view.cpu().numpy()) == want
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
        try:
            sizes, gathered = sharding.all_gather_sections(view)
            lf_sizes, lf_gathered = sharding.all_gather_sections(ctx.lf_payload_tensor())
            # the context may reuse its buffers only after the collectives have read them (bench.py exchange())
            sharding.fence_context_stream(ctx)
            ctx.encode_image_tensor(t)       # next frame on the same context, queued behind the fence
            ctx.sync()
            torch.cuda.synchronize()
            assert [int(x) for x in sizes] == [len(want)]
            assert sharding.concatenate(sizes, gathered) == want
            assert int(lf_sizes[0]) == ctx.lf_payload_size() > 0
            assert ctx.read_payload() == want
        finally:
            dist.destroy_process_group()


def _nccl_world_of_one():
    import socket

    import torch
    import torch.distributed as dist

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    return dist


@pytest.mark.parametrize("kind,w,h,depth", [("photo", 4096 + 200, 2048 + 72, 8), ("photo", 2048 + 64, 2048 + 64, 32)])
def test_encode_distributed_over_rccl_world_of_one(image, kind, w, h, depth):
    """The one-process-per-GPU entry point itself (multigpu.encode_distributed), over the "nccl" (= RCCL)
    backend in a world of one: transform -> all-gather of maxima on the device -> entropy stage with
    the floor read from device memory -> hydamd_export_frame -> gather -> hydamd_frame_from_blobs."""
    import torch
    from hydrium_amd import multigpu, synth
    from oracle import refprobe

    if depth == 32:
        img = synth.make_image_f32(kind, w, h)
        img[:64, :64] *= 300.0  # out-of-gamut floats: alphabets above 32
    else:
        img = image(kind, w, h, depth)
    t = _cuda(img)
    dist = _nccl_world_of_one()
    try:
        origin = lambda lf: ((lf // (-(-w // 2048))) * 2048, (lf % (-(-w // 2048))) * 2048)
        got = multigpu.encode_distributed(t, w, h, origin, exchange=True)  # the collectives, although nobody else is there
        torch.cuda.synchronize()
        # the same frame put together by rank 0's host from the gathered blobs (round 2's path), and with the exchange
        # left out as a lone rank does by default
        assert multigpu.encode_distributed(t, w, h, origin, exchange=True, assemble_on_device=False) == got
        assert multigpu.encode_distributed(t, w, h, origin) == got
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    assert got == api.encode_image(api.Library(), np.ascontiguousarray(img))
    if reference_expected():
        assert got == api.encode_image(refprobe.reference_library(), np.ascontiguousarray(img))


@pytest.mark.parametrize("shards", [2, 5])
def test_blobs_of_several_shards_make_the_reference_file(image, shards):
    """hydamd_export_frame / hydamd_frame_from_blobs with the LF groups of one frame spread over several
    contexts (what N ranks hold after the gather), floors handed over through device memory."""
    import torch
    from hydrium_amd import device, multigpu, sharding

    img = image("photo", 4096 + 200, 2 * 2048 + 72, 8)
    h, w, _ = img.shape
    lfx = -(-w // 2048)
    t = _cuda(img)
    parts = sharding.partition_lf_groups(lfx * (-(-h // 2048)), shards)
    engines = [multigpu.GpuShardEngine(multigpu.Shard(0, p, w, h), t, lambda lf: ((lf // lfx) * 2048, (lf % lfx) * 2048))
               for p in parts]
    try:
        for e in engines:
            e.enqueue_transform()
        torch.cuda.synchronize()
        maxima = [int(v) for e in engines for v in e.alphabet_maxima().cpu()]
        blobs, seen = [], 0
        for e in engines:
            floor = torch.tensor([max(maxima[:seen], default=0)], dtype=torch.int32, device="cuda")
            seen += e.n
            e.enqueue_entropy(floor)
            out = torch.zeros(e.blob_bound(), dtype=torch.uint8, device="cuda")
            e.export_blob(out)
            e.finish()
            head = device.blob_header(out[:64].cpu().numpy().tobytes())
            assert int(head["status"]) == 0 and int(head["num_slots"]) == e.n
            blobs.append(out[:int(head["total_bytes"])].cpu().numpy().tobytes())
    finally:
        for e in engines:
            e.shard.close()
    got = device.frame_from_blobs(api.HYDImageMetadata(w, h, 0, -1, -1), blobs)
    assert got == api.encode_image(api.Library(), img)
