"""CPU, world_size 2 over gloo: the N>1 path of bench.py / multi-GPU encode — LF-group partitioning
and the all-gather that concatenates every rank's coded sections — without any GPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from hydrium_amd import sharding


def test_partition_covers_every_lf_group_once():
    for n in (1, 4, 15, 16, 64, 255):
        for world in (1, 2, 3, 8):
            parts = sharding.partition_lf_groups(n, world)
            assert len(parts) == world
            flat = [i for p in parts for i in p]
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_slab_grid_shapes():
    assert sharding.slab_grid(1) == (1, 1)
    assert sharding.slab_grid(2) == (2, 1)
    assert sharding.slab_grid(4) == (2, 2)
    assert sharding.slab_grid(8) == (4, 2)
    for w in range(1, 17):
        gx, gy = sharding.slab_grid(w)
        assert gx * gy == w


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # every rank "codes" its own LF groups with the CPU oracle, then the sections are gathered
        from hydrium_amd import synth
        from oracle import binding as orc

        img = synth.make_image("photo", 2048 + 300, 40, 8)  # 2 LF groups side by side
        mine = sharding.partition_lf_groups(2, world)[rank]
        mx, chunks = 0, []
        for lf in range(2):
            res, mx = orc.encode_lf_group(img, lf, 0, max_alphabet_size=mx)  # running alphabet needs send order
            if lf in mine:
                chunks.append(res.stream)
        payload = torch.from_numpy(np.frombuffer(b"".join(chunks), np.uint8).copy())
        sizes, gathered = sharding.all_gather_sections(payload)
        exact = sharding.concatenate(sizes, gathered)
        # the streaming variant: a capacity every rank derives from the sizes it was just told, one
        # collective per call, staging buffers reused
        cap = int(int(sizes.max()) * 1.25) + 4096
        bufs = {}
        for _ in range(2):
            s2, g2 = sharding.all_gather_sections(payload, capacity=cap, buffers=bufs)
            assert [int(x) for x in s2] == [int(x) for x in sizes]
            assert sharding.concatenate(s2, g2) == exact
        # and the point-to-point shape: everything to the assembling rank only
        s3, g3 = sharding.gather_sections(payload, cap, dst=0, buffers=bufs)
        if rank == 0:
            assert [int(x) for x in s3] == [int(x) for x in sizes]
            assert sharding.concatenate(s3, g3) == exact
        else:
            assert s3 is None and g3 is None
        q.put((rank, [int(x) for x in sizes], exact))
    finally:
        dist.destroy_process_group()


def test_all_gather_sections_world2_matches_single_process():
    from hydrium_amd import synth
    from oracle import binding as orc

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    img = synth.make_image("photo", 2048 + 300, 40, 8)
    mx, want = 0, b""
    for lf in range(2):
        res, mx = orc.encode_lf_group(img, lf, 0, max_alphabet_size=mx)
        want += res.stream
    for rank, sizes, data in results:
        assert sum(sizes) == len(want)
        assert data == want, f"rank {rank} assembled different bytes"


def _choreography_worker(rank, world, port, q, shape, retry_rank):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys

        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import glue
        import oracle_engine
        from hydrium_amd import api, device, multigpu, synth

        w, h, depth = shape
        img = synth.make_image_f32("photo", w, h) if depth == 32 else synth.make_image("photo", w, h, depth)
        n_lf = (-(-w // 2048)) * (-(-h // 2048))
        parts = sharding.partition_lf_groups(n_lf, world)
        engine = oracle_engine.OracleShardEngine(img, parts[rank], fail_first_export=rank == retry_rank)
        blobs = multigpu.choreograph_frame(engine, parts)
        data = None
        if rank == 0:
            if glue._d is None:
                glue._d = glue._lib()
            import ctypes as C

            d = glue._d  # the product's host glue without a GPU (libhydrium_hosttest.so)
            d.hydamd_frame_from_blobs.restype = C.c_int
            d.hydamd_frame_from_blobs.argtypes = [C.POINTER(api.HYDImageMetadata), C.c_int, C.c_int, C.c_size_t,
                                                  C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t,
                                                  C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_char_p)]
            d.hydamd_free.argtypes = [C.c_void_p]
            data = device.frame_from_blobs(api.HYDImageMetadata(w, h, 0, -1, -1), blobs, lib=d)
        q.put((rank, data, engine.exports))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("shape,retry_rank", [((2048 + 300, 2048 + 40, 8), -1), ((2048 * 2 + 100, 300, 32), -1),
                                              ((2048 + 300, 40, 8), 1)])
def test_sharded_frame_choreography_world2(ref_lib, shape, retry_rank):
    """multigpu.choreograph_frame in two processes over gloo, shards coded by the CPU oracle: the frame
    rank 0 assembles from the gathered blobs is the reference's file.  The float image's alphabet grows
    from LF group to LF group, so the floor exchange matters (entropy.c:459-460); `retry_rank` reports
    one incomplete blob, which makes every rank run the frame again."""
    from hydrium_amd import api, synth

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_choreography_worker, args=(r, world, port, q, shape, retry_rank)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict((r[0], r[1:]) for r in (q.get(timeout=300) for _ in range(world)))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    w, h, depth = shape
    img = synth.make_image_f32("photo", w, h) if depth == 32 else synth.make_image("photo", w, h, depth)
    assert results[0][0] == api.encode_image(ref_lib, img)
    assert results[1][0] is None
    assert all(results[r][1] == (2 if retry_rank >= 0 else 1) for r in range(world))
