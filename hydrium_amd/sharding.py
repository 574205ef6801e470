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


def all_gather_sections(payload, group=None):
    """Concatenate every rank's packed sections.

    `payload` is this rank's 1-D uint8 tensor (any length).  Returns (sizes, gathered) where
    `sizes[r]` is rank r's byte count and `gathered[r, :sizes[r]]` its bytes, on every rank.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    n = torch.tensor([payload.numel()], dtype=torch.int64, device=payload.device)
    sizes = torch.empty(world, dtype=torch.int64, device=payload.device)
    dist.all_gather_into_tensor(sizes, n, group=group)
    cap = int(sizes.max().item())
    mine = torch.zeros(max(cap, 1), dtype=torch.uint8, device=payload.device)
    mine[:payload.numel()] = payload
    gathered = torch.empty(world * max(cap, 1), dtype=torch.uint8, device=payload.device)
    dist.all_gather_into_tensor(gathered, mine, group=group)
    return sizes, gathered.view(world, max(cap, 1))


def fence_context_stream(ctx) -> None:
    """Make everything queued on the context's stream from now on wait for what torch's current
    stream holds at this point — the collectives that are still reading the context's payload
    buffers.  (Contexts run on their own HIP streams; torch knows nothing about them.)"""
    import torch

    ev = torch.cuda.Event()
    ev.record()
    ev.wait(torch.cuda.ExternalStream(ctx.get_stream()))


def concatenate(sizes: Sequence[int], gathered) -> bytes:
    """Host-side helper: the rank-ordered byte string of all sections."""
    return b"".join(bytes(gathered[r, :int(sizes[r])].cpu().numpy()) for r in range(len(sizes)))
