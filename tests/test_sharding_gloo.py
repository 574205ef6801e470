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
