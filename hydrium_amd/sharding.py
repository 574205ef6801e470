"""Multi-GPU plumbing: how a picture is split across ranks and how coded sections are exchanged.

The hot path shards by LF group (2048x2048): histograms, ANS tables and the LF plane are all
LF-group-local (reference encoder.c:852,928-939), so ranks never exchange anything until their
HF sections are final.  The only collective is the concatenation of those sections
(SURVEY.md §8e): one all-gather of byte counts, one of the (padded) byte payloads.  Works with
any torch.distributed backend: "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple


def slab_grid(world: int) -> Tuple[int, int]:
    """Arrange `world` equally sized slabs as close to a square as possible: (across, down)."""
    gx = 1
    while gx * gx < world:
        gx *= 2
    gx = min(gx, world)
    while world % gx:
        gx -= 1
    return gx, world // gx


def partition_lf_groups(num_lf_groups: int, world: int) -> List[range]:
    """Contiguous raster-order blocks of LF groups per rank; earlier ranks take the remainder."""
    base, extra = divmod(num_lf_groups, world)
    out, start = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append(range(start, start + n))
        start += n
    return out


def all_gather_sections(payload, group=None, capacity=None, buffers=None):
    """Concatenate every rank's packed sections.

    `payload` is this rank's 1-D uint8 tensor (any length).  Returns (sizes, gathered) where
    `sizes[r]` is rank r's byte count and `gathered[r, :sizes[r]]` its bytes, on every rank.

    The collective needs one common per-rank capacity.  By default it is the largest payload, which
    costs a host synchronisation (the sizes have to come back before the second collective can be
    shaped).  A caller that streams many frames of similar size passes `capacity` — a bound every
    rank agrees on, e.g. 1.25 x the largest size of the first frames, which all ranks know from the
    `sizes` they were returned — and the call stays asynchronous; `buffers` (a dict the caller
    keeps) lets it reuse its staging tensors instead of allocating per frame.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    dev = payload.device
    if capacity is not None:
        # agreed capacity: ONE collective per frame, each rank's byte count rides in front of its bytes
        cap = (max(int(capacity), 1) + 7) & ~7  # rows stay 8-byte aligned for the int64 size header
        if payload.numel() > cap:
            raise ValueError(f"payload of {payload.numel()} bytes exceeds the agreed capacity {cap}")
        pitch = 8 + cap
        if buffers is not None and buffers.get("pitch") == pitch and buffers["mine"].device == dev:
            mine, gathered = buffers["mine"], buffers["gathered"]
        else:
            mine = torch.zeros(pitch, dtype=torch.uint8, device=dev)
            gathered = torch.empty(world * pitch, dtype=torch.uint8, device=dev)
            if buffers is not None:
                buffers.update(pitch=pitch, mine=mine, gathered=gathered)
        mine[:8].view(torch.int64).fill_(payload.numel())
        mine[8:8 + payload.numel()] = payload  # bytes past the payload are whatever an earlier frame left
        dist.all_gather_into_tensor(gathered, mine, group=group)
        rows = gathered.view(world, pitch)
        return rows[:, :8].clone().view(torch.int64).view(world), rows[:, 8:]
    n = torch.tensor([payload.numel()], dtype=torch.int64, device=dev)
    sizes = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes, n, group=group)
    cap = max(int(sizes.max().item()), 1)
    mine = torch.zeros(cap, dtype=torch.uint8, device=dev)
    mine[:payload.numel()] = payload
    gathered = torch.empty(world * cap, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(gathered, mine, group=group)
    return sizes, gathered.view(world, cap)


def gather_sections(payload, capacity, dst=0, group=None, buffers=None):
    """Bring every rank's packed sections to ONE rank — the one that assembles the frame.

    On a point-to-point fabric (xGMI: one link per peer) this is the cheap shape of the exchange: each
    rank sends its ~10 MB once, the assembling rank receives world - 1 chunks over world - 1 different
    links, and nobody receives data it has no use for (an all-gather moves world times as much).
    `capacity` is the agreed per-rank bound (see all_gather_sections); each rank's byte count rides in
    front of its bytes.  Returns (sizes, rows) on rank `dst`, (None, None) elsewhere.
    """
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = payload.device
    cap = (max(int(capacity), 1) + 7) & ~7
    if payload.numel() > cap:
        raise ValueError(f"payload of {payload.numel()} bytes exceeds the agreed capacity {cap}")
    pitch = 8 + cap
    if buffers is not None and buffers.get("gpitch") == pitch and buffers["gmine"].device == dev:
        mine, gathered = buffers["gmine"], buffers["ggathered"]
    else:
        mine = torch.zeros(pitch, dtype=torch.uint8, device=dev)
        gathered = torch.empty(world * pitch, dtype=torch.uint8, device=dev) if rank == dst else None
        if buffers is not None:
            buffers.update(gpitch=pitch, gmine=mine, ggathered=gathered)
    mine[:8].view(torch.int64).fill_(payload.numel())
    mine[8:8 + payload.numel()] = payload
    if rank == dst:
        rows = gathered.view(world, pitch)
        dist.gather(mine, gather_list=[rows[r] for r in range(world)], dst=dst, group=group)
        return rows[:, :8].clone().view(torch.int64).view(world), rows[:, 8:]
    dist.gather(mine, gather_list=None, dst=dst, group=group)
    return None, None


def fence_context_stream(ctx) -> None:
    """Make everything queued on the context's stream from now on wait for what torch's current
    stream holds at this point — the collectives that are still reading the context's payload
    buffers.  (Contexts run on their own HIP streams; torch knows nothing about them.)"""
    import torch

    ext = getattr(ctx, "_torch_stream", None)
    if ext is None or ext.cuda_stream != ctx.get_stream():
        ext = ctx._torch_stream = torch.cuda.ExternalStream(ctx.get_stream())
    ev = torch.cuda.Event()
    ev.record()
    ev.wait(ext)


def concatenate(sizes: Sequence[int], gathered) -> bytes:
    """Host-side helper: the rank-ordered byte string of all sections."""
    return b"".join(bytes(gathered[r, :int(sizes[r])].cpu().numpy()) for r in range(len(sizes)))
