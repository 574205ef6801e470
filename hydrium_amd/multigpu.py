"""Sharded encode of ONE frame over several GPUs (or several contexts): BASELINE config C4.

Each shard owns a contiguous raster-order block of LF groups (``sharding.partition_lf_groups``),
keeps its part of the picture in its own HBM and runs the hot path on it.  Two things cross
shards, both tiny next to the pixels:

* before the entropy stage, one integer per LF group (largest token + 1) so that every shard can
  set its running-alphabet floor (reference entropy.c:459-460: the maximum is never reset between
  LF groups);
* after it, the shard's results: packed HF sections (the all-gather of ``sharding``), the LF
  coefficient streams coded by the GPU LF coder (or LF ints when it is off), frequency tables and
  section sizes.  Rank 0 wraps them with ``hydamd_frame_from_streams`` / ``hydamd_frame_from_results``.

``Shard`` holds one shard's state; ``encode_serial`` drives N shards one after another in a single
process (that is what the single-GPU tests run), ``encode_distributed`` is the same choreography
with ``torch.distributed`` collectives, one process per GPU.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np

from . import api, device, sharding


class Shard:
    def __init__(self, dev_index: int, lf_ids: Sequence[int], width: int, height: int, linear_light: int = 0):
        self.lf_ids = list(lf_ids)
        self.width, self.height = width, height
        self.lfx = -(-width // 2048)
        self.lfy = -(-height // 2048)
        self.ctx = device.DeviceContext(dev_index, max(1, len(self.lf_ids)), linear_light) if self.lf_ids else None

    def geometry(self, lf: int):
        tx, ty = lf % self.lfx, lf // self.lfx
        return tx, ty, min(2048, self.width - tx * 2048), min(2048, self.height - ty * 2048)

    def submit(self, tensor, origin_lf_pixels):
        """Queue this shard's LF groups.  ``tensor`` is an interleaved (h, w, 3) CUDA tensor holding at
        least the shard's pixels; ``origin_lf_pixels(lf) -> (row, col)`` gives where LF group ``lf`` starts in it."""
        if not self.lf_ids:
            return
        isz = tensor.element_size()
        fmt = {1: 0, 2: 1, 4: 2}[isz]
        pitch = tensor.shape[1]
        self.ctx.begin_frame(self.lfx * self.lfy)
        for slot, lf in enumerate(self.lf_ids):
            _, _, w, h = self.geometry(lf)
            r, c = origin_lf_pixels(lf)
            p = tensor.data_ptr() + (r * pitch + c) * 3 * isz
            self.ctx.encode_lf_group(slot, [p, p + isz, p + 2 * isz], 3 * pitch, 3, fmt, w, h, lf)
        self.ctx.run_transform(len(self.lf_ids))

    def alphabet_maxima(self) -> List[int]:
        return [self.ctx.read_alphabet_max(s) for s in range(len(self.lf_ids))] if self.lf_ids else []

    def entropy(self, floor: int):
        if not self.lf_ids:
            return
        self.ctx.set_alphabet_floor(floor)
        self.ctx.run_entropy(len(self.lf_ids))

    def results(self):
        """dict of this shard's results, in its LF-group order (host copies)."""
        out = dict(tiles=[], dc=[], lf=[], freq=[], alphabet=[], bits=[], payload=b"", running_max=0)
        if not self.lf_ids:
            return out
        self.ctx.sync()
        out["payload"] = self.ctx.read_payload()
        for slot, lf in enumerate(self.lf_ids):
            tx, ty, w, h = self.geometry(lf)
            freq, alpha, _, running = self.ctx.read_tables(slot)
            bits, _ = self.ctx.read_sections(slot)
            out["tiles"].append((tx, ty))
            if self.ctx.lf_coder():  # LF coefficients were coded on this GPU: ship the stream, not the ints
                lengths, alpha_lf, pairs, nbits = self.ctx.read_lf_stream(slot)
                out["lf"].append((lengths, alpha_lf, pairs, nbits, self.ctx.read_lf_bits(slot, nbits)))
            else:
                out["dc"].append(self.ctx.read_dc(slot, -(-w // 8), -(-h // 8)))
            out["freq"].append(freq)
            out["alphabet"].append(alpha)
            out["bits"].append(bits)
            out["running_max"] = max(out["running_max"], running)
        return out

    def close(self):
        if self.ctx:
            self.ctx.close()


def assemble(width: int, height: int, shard_results: Sequence[dict], linear_light: int = 0,
             icc: Optional[bytes] = None) -> bytes:
    md = api.HYDImageMetadata(width, height, linear_light, -1, -1)
    tiles = [t for r in shard_results for t in r["tiles"]]
    dcs = [a for r in shard_results for a in r["dc"]]
    lfs = [a for r in shard_results for a in r.get("lf", [])]
    if lfs and dcs:
        raise ValueError("shards must agree on where the LF coefficients are coded")
    return device.frame_from_results(
        md, tiles, dcs, [a for r in shard_results for a in r["freq"]],
        [a for r in shard_results for a in r["alphabet"]], [a for r in shard_results for a in r["bits"]],
        max(r["running_max"] for r in shard_results), b"".join(r["payload"] for r in shard_results), icc=icc,
        lf_streams=lfs if lfs else None)


def encode_serial(img_tensor, num_shards: int, dev_index: int = 0, linear_light: int = 0) -> bytes:
    """N-way sharded encode of a whole (H, W, 3) CUDA tensor, shards run one after another here."""
    h, w, _ = img_tensor.shape
    lfx = -(-w // 2048)
    n_lf = lfx * (-(-h // 2048))
    parts = sharding.partition_lf_groups(n_lf, num_shards)
    shards = [Shard(dev_index, part, w, h, linear_light) for part in parts]
    try:
        for s in shards:
            s.submit(img_tensor, lambda lf: ((lf // lfx) * 2048, (lf % lfx) * 2048))
        maxima = [m for s in shards for m in s.alphabet_maxima()]  # raster LF-group order
        seen = 0
        for s in shards:
            s.entropy(max(maxima[:seen], default=0))
            seen += len(s.lf_ids)
        return assemble(w, h, [s.results() for s in shards], linear_light)
    finally:
        for s in shards:
            s.close()


def encode_distributed(slab_tensor, width: int, height: int, origin_lf_pixels, group=None, linear_light: int = 0):
    """One process per GPU: this rank codes its LF groups out of ``slab_tensor`` (its part of the
    picture, already in its HBM); rank 0 returns the codestream, the others ``None``."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lfx = -(-width // 2048)
    n_lf = lfx * (-(-height // 2048))
    parts = sharding.partition_lf_groups(n_lf, world)
    shard = Shard(torch.cuda.current_device(), parts[rank], width, height, linear_light)
    try:
        shard.submit(slab_tensor, origin_lf_pixels)
        all_max: List[List[int]] = [None] * world
        dist.all_gather_object(all_max, shard.alphabet_maxima(), group=group)
        before = [m for r in range(rank) for m in all_max[r]]
        shard.entropy(max(before, default=0))
        res = shard.results()
        payload = res.pop("payload")
        dev = slab_tensor.device
        sizes, gathered = sharding.all_gather_sections(
            torch.from_numpy(np.frombuffer(payload, np.uint8).copy()).to(dev), group)
        metas: List[dict] = [None] * world
        dist.all_gather_object(metas, res, group=group)
        if rank != 0:
            return None
        host = gathered.cpu().numpy()
        for r in range(world):
            metas[r]["payload"] = host[r, :int(sizes[r])].tobytes()
        return assemble(width, height, metas, linear_light)
    finally:
        shard.close()
