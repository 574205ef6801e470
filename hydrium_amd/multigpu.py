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
process (that is what the single-GPU tests run).  ``encode_distributed`` is the one-process-per-GPU
form: ``choreograph_frame`` keeps everything on the devices — the maxima are all-gathered as a
tensor, each shard leaves ONE blob (``hydamd_export_frame``: tables, section sizes, LF streams, packed
sections) that is gathered to the assembling rank, whose host builds the frame with
``hydamd_frame_from_blobs``.
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


class GpuShardEngine:
    """What ``choreograph_frame`` needs from a shard, on a GPU: every call only ENQUEUES work on the
    shard context's HIP stream (made torch's current stream, so that torch ops and RCCL collectives
    order themselves behind the kernels without explicit events); ``finish`` is the one host wait."""

    def __init__(self, shard: Shard, slab_tensor, origin_lf_pixels):
        import torch

        self.shard, self.tensor, self.origin = shard, slab_tensor, origin_lf_pixels
        self.n = len(shard.lf_ids)
        self.device = slab_tensor.device
        self.stream = torch.cuda.ExternalStream(shard.ctx.get_stream()) if shard.ctx else torch.cuda.current_stream()

    def enqueue_transform(self):
        self.shard.submit(self.tensor, self.origin)

    def alphabet_maxima(self):
        import torch

        if not self.n:
            return torch.zeros(0, dtype=torch.int32, device=self.device)
        return self.shard.ctx.alphabet_max_tensor(self.n)

    def enqueue_entropy(self, floor_tensor):
        if self.n:
            if floor_tensor is not None:  # None: no LF group of the frame is coded before this shard's
                self.shard.ctx.set_alphabet_floor_device(floor_tensor)
            self.shard.ctx.run_entropy(self.n)

    def blob_bound(self) -> int:
        return self.shard.ctx.blob_bound(self.n) if self.n else device.BLOB_HEADER_DTYPE.itemsize

    def export_blob(self, out):
        if self.n:
            self.shard.ctx.export_frame(self.n, out)
        else:  # a rank without LF groups sends an empty blob
            import torch

            h = np.zeros(1, device.BLOB_HEADER_DTYPE)
            h["magic"], h["version"], h["lf_coded"] = device.BLOB_MAGIC, 1, 1
            h["total_bytes"] = device.BLOB_HEADER_DTYPE.itemsize
            out[:h.nbytes] = torch.from_numpy(h.view(np.uint8).copy()).to(out.device)

    def finish(self):
        """Wait for the shard; a frame that outgrew the context's buffers is rerun in here (hydamd_sync)."""
        if self.n:
            self.shard.ctx.sync()
        else:
            self.stream.synchronize()

    def overflow_reruns(self) -> int:
        return self.shard.ctx.overflow_reruns() if self.n else 0

    def alphabet_device(self):
        return self.device


class FrameAssembly:
    """What the assembling rank needs to build frames where the blobs are (device.Assembler): the
    assembler, the frame's description and an output buffer — pinned host memory by default, so that the
    finished codestream lands on the host with the assembly itself and no copy follows."""

    def __init__(self, dev_index: int, width: int, height: int, capacity: Optional[int] = None, linear_light: int = 0,
                 pinned: bool = True, icc: Optional[bytes] = None):
        self.asm = device.Assembler(dev_index)
        self.md = api.HYDImageMetadata(width, height, linear_light, -1, -1)
        self.icc = icc
        self.dev_index, self.pinned = dev_index, pinned
        self.out = None
        if capacity:
            self._allocate(capacity)
        self.size = 0

    def _allocate(self, capacity: int):
        import torch

        capacity = (capacity + 255) & ~255
        self.out = (torch.empty(capacity, dtype=torch.uint8).pin_memory() if self.pinned
                    else torch.empty(capacity, dtype=torch.uint8, device=torch.device("cuda", self.dev_index)))
        # the landing buffer now, not at the first copy: pinning 64 MB takes tens of milliseconds (with eight frames in
        # flight, five of the eight first copies used to fall inside bench.py's timed interval)
        self._host = None if self.pinned else torch.empty(capacity, dtype=torch.uint8).pin_memory()

    @staticmethod
    def supports(width: int, height: int) -> bool:
        """Frames of a single 256 x 256 group are one bit-contiguous section: the host assembler takes those."""
        return (-(-width // 256)) * (-(-height // 256)) > 1

    def enqueue(self, rows, parts):
        """On torch's current stream, behind whatever fills `rows` (the gathered blobs, rank order)."""
        keep = [(r, p) for r, p in zip(rows, parts) if len(p)]  # a rank without LF groups sends an empty blob
        need = sum(r.numel() for r, _ in keep) + (1 << 20)     # no frame is larger than its blobs plus its headers
        if self.out is None or self.out.numel() < need:
            self._allocate(need)
        if getattr(self, "_copy_stream", None) is not None:
            # the copy of the previous frame out of this buffer (start_copy) may still be running: the kernels that
            # overwrite it wait for it on the device
            import torch

            torch.cuda.current_stream().wait_event(self._copied)
        self.asm.plan(self.md, [list(p) for _, p in keep], icc=self.icc)
        self.asm.run_tensors([r for r, _ in keep], self.out)

    def finish(self):
        """After the stream has been synchronised: the frame as a uint8 view of the output buffer (no copy when pinned)."""
        self.size = self.asm.result()
        return self.out[:self.size]

    _dbg = []
    _copy_streams = {}  # device index -> the one D2H stream of that device

    def start_copy(self):
        """Device output: start the one D2H copy of the finished frame (the frame is complete: finish() came after the
        stream's synchronisation), so that it overlaps the next frames' kernels.
        * The copy's length is rounded up to 256 bytes (the buffers have the room): a length that is not a multiple of
          four sends the runtime down its shader-copy path (__amd_rocclr_copyBuffer in 3 MB pieces), which gets
          2-7 GB/s beside the transform kernels; the DMA engines do 56 GB/s whatever the CUs do (scripts/d2h_probe.py,
          scripts/ubench/d2h_kernel.hip).
        * All assemblies of a device share ONE copy stream — the copies follow each other on the link anyway, a stream
          per frame in flight takes a hardware queue each — and each has an event of its own to wait for."""
        import os

        import torch

        if self.pinned or os.environ.get("HYDAMD_BENCH_SKIP_D2H"):  # (the switch exists to measure what the copy costs)
            return
        if getattr(self, "_host", None) is None or self._host.numel() < self.out.numel():
            self._host = torch.empty(self.out.numel(), dtype=torch.uint8).pin_memory()
        if getattr(self, "_copy_stream", None) is None:
            if self.dev_index not in FrameAssembly._copy_streams:
                FrameAssembly._copy_streams[self.dev_index] = torch.cuda.Stream(device=torch.device("cuda", self.dev_index))
            self._copy_stream = FrameAssembly._copy_streams[self.dev_index]
            self._copied = torch.cuda.Event()
        n = min(self.out.numel() & ~255, (self.size + 255) & ~255)
        dbg = os.environ.get("HYDAMD_DEBUG_D2H")
        if dbg:
            import time
            t0 = time.perf_counter()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(self._copy_stream):
            if dbg:
                e0.record(self._copy_stream)
            self._host[:n].copy_(self.out[:n], non_blocking=True)
            if dbg:
                e1.record(self._copy_stream)
            self._copied.record(self._copy_stream)
        if dbg:
            FrameAssembly._dbg.append((time.perf_counter() - t0, e0, e1, n))
        self._copying = True

    def to_host(self):
        """The finished frame in pinned host memory: the output buffer itself, or the copy start_copy() began
        (started here if it was not)."""
        import os

        if self.pinned:
            return self.out[:self.size]
        if not getattr(self, "_copying", False):
            self.start_copy()
        if getattr(self, "_copy_stream", None) is None:
            return self.out[:self.size]
        if os.environ.get("HYDAMD_DEBUG_D2H"):
            import time
            t0 = time.perf_counter()
            self._copied.synchronize()
            FrameAssembly._dbg.append((time.perf_counter() - t0,))
        self._copied.synchronize()
        self._copying = False
        return self._host[:self.size]

    def close(self):
        self.asm.close()
        self.out = self._host = self._copy_stream = self._copied = None  # device blocks torch handed out under a context's stream go back before that stream dies


def enqueue_frame(engine, parts, capacity: int, group=None, to_rank=0, assembly: Optional["FrameAssembly"] = None,
                  exchange: Optional[bool] = None):
    """Everything a frame needs from this rank, enqueued without a host wait: transform stage ->
    all-gather of the per-LF-group alphabet maxima (one int32 per LF group, on the device) -> entropy
    stage with this rank's floor -> the shard's blob -> one gather of the blobs to ``to_rank`` -> with an
    ``assembly``, the frame itself, built on ``to_rank``'s GPU from the gathered blobs (device.Assembler).
    ``capacity`` is the blob size every rank sends (all ranks must pass the same value).  Returns a
    handle for ``collect_frame``.  Reference: presets are numbered across the whole frame
    (encoder.c:852-901) and the alphabet maximum runs over LF groups in send order
    (entropy.c:459-460,952).  ``exchange``: run the two collectives even in a world of one (None: only
    when there is someone to exchange with — a lone rank's floor is zero and its blob already is where
    the assembler reads it)."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    collectives = world > 1 if exchange is None else bool(exchange)
    most = max(max(len(p) for p in parts), 1)
    reruns = engine.overflow_reruns() if hasattr(engine, "overflow_reruns") else None
    engine.enqueue_transform()
    mine = engine.alphabet_maxima()
    every = padded = None
    if collectives:
        padded = torch.zeros(most, dtype=torch.int32, device=mine.device)
        padded[:mine.numel()] = mine
        every = torch.empty(world * most, dtype=torch.int32, device=mine.device)
        dist.all_gather_into_tensor(every, padded, group=group)
        before = every.view(world, -1)[:rank]
        floor = (before.max() if before.numel() else torch.zeros((), dtype=torch.int32, device=mine.device)).reshape(1)
        floor = floor.to(torch.int32).contiguous()
    else:
        floor = None  # no LF group is coded before this rank's
    engine.enqueue_entropy(floor)
    cap = (int(capacity) + 15) & ~15
    blob = torch.empty(cap, dtype=torch.uint8, device=mine.device)
    engine.export_blob(blob)
    if collectives:
        rows = [torch.empty(cap, dtype=torch.uint8, device=mine.device) for _ in range(world)] if rank == to_rank else None
        dist.gather(blob, gather_list=rows, dst=to_rank, group=group)
    else:
        rows = [blob]
    if assembly is not None and rank == to_rank:
        assembly.enqueue(rows, parts)
    return dict(engine=engine, blob=blob, rows=rows, to_rank=to_rank, rank=rank, keep=(every, padded, floor),
                assembly=assembly if rank == to_rank else None, reruns=reruns)


def collect_frame(handle):
    """Wait for this rank's part of a frame queued by ``enqueue_frame``.  Returns (retry, rows): ``retry``
    says a blob is incomplete (a shard's frame outgrew a buffer and was rerun inside the wait, or its
    blob did not fit ``capacity``) — every rank then has to run the frame again; ``rows`` are the
    gathered blobs as device tensors on the assembling rank, None elsewhere.  With an assembly the frame
    is in ``handle["frame"]`` afterwards (a view of the assembly's output buffer) and nothing is read back
    here: this rank's own rerun shows in the context's counter, anyone's incomplete blob in the
    assembler's verdict."""
    engine = handle["engine"]
    engine.finish()
    asm = handle.get("assembly")
    if asm is not None:
        try:
            handle["frame"] = asm.finish()
            return False, handle["rows"]
        except device.DeviceError as e:
            if "incomplete" in e.message:
                return True, handle["rows"]
            raise
    if handle.get("reruns") is not None and handle["rank"] != handle["to_rank"]:
        return engine.overflow_reruns() != handle["reruns"], handle["rows"]
    head = device.blob_header(handle["blob"][:device.BLOB_HEADER_DTYPE.itemsize].cpu().numpy().tobytes())
    return bool(int(head["status"]) & device.BLOB_RETRY), handle["rows"]


def choreograph_frame(engine, parts, group=None, capacity=None, to_rank=0, assembly: Optional["FrameAssembly"] = None,
                      exchange: Optional[bool] = None):
    """One frame over the ranks of ``group`` (enqueue_frame + collect_frame, rerun if a shard says so).
    Returns the list of blobs (bytes, rank order) on ``to_rank`` — or, with an ``assembly``, the frame
    itself as bytes — and ``None`` elsewhere.  ``engine`` is a GpuShardEngine or anything with its
    methods; ``parts`` the LF-group partition (sharding.partition_lf_groups)."""
    import torch
    import torch.distributed as dist

    rank = dist.get_rank(group)
    dev = engine.alphabet_device() if hasattr(engine, "alphabet_device") else None
    for attempt in range(3):
        # every rank sends the same number of bytes: the largest bound any of them reports
        cap = torch.tensor([capacity or engine.blob_bound()], dtype=torch.int64, device=dev)
        dist.all_reduce(cap, op=dist.ReduceOp.MAX, group=group)
        handle = enqueue_frame(engine, parts, int(cap.item()), group, to_rank, assembly, exchange)
        again, rows = collect_frame(handle)
        retry = torch.tensor([1 if again else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(retry, op=dist.ReduceOp.MAX, group=group)
        if not int(retry.item()):
            if rank != to_rank:
                return None
            if assembly is not None:
                return bytes(handle["frame"].cpu().numpy())
            out = []
            for r in rows:
                host = r.cpu().numpy()
                out.append(host[:int(device.blob_header(host.tobytes()[:64])["total_bytes"])].tobytes())
            return out
        capacity = None  # the rerun enlarged the context: take its new bound
    raise RuntimeError("frame did not fit its buffers after two enlargements")


def encode_distributed(slab_tensor, width: int, height: int, origin_lf_pixels, group=None, linear_light: int = 0,
                       lib=None, assemble_on_device: bool = True, exchange: Optional[bool] = None):
    """One process per GPU: this rank codes its LF groups out of ``slab_tensor`` (its part of the
    picture, already in its HBM); rank 0 returns the codestream, the others ``None``.  The frame is put
    together on rank 0's GPU from the gathered blobs and written straight into pinned host memory
    (``assemble_on_device=False``: the blobs cross to the host and hydamd_frame_from_blobs builds it there,
    which is also what a frame of a single group gets)."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lfx = -(-width // 2048)
    n_lf = lfx * (-(-height // 2048))
    parts = sharding.partition_lf_groups(n_lf, world)
    shard = Shard(torch.cuda.current_device(), parts[rank], width, height, linear_light)
    assembly = None
    try:
        engine = GpuShardEngine(shard, slab_tensor, origin_lf_pixels)
        if assemble_on_device and rank == 0 and FrameAssembly.supports(width, height):
            assembly = FrameAssembly(torch.cuda.current_device(), width, height, None, linear_light)
        with torch.cuda.stream(engine.stream):
            blobs = choreograph_frame(engine, parts, group, assembly=assembly, exchange=exchange)
        if blobs is None or assembly is not None:
            return blobs
        return device.frame_from_blobs(api.HYDImageMetadata(width, height, linear_light, -1, -1), blobs, lib=lib)
    finally:
        if assembly is not None:
            assembly.close()
        # torch's caching allocator pools the blocks it handed out under the context's stream by that
        # stream: give them back to the driver before the stream is destroyed with the context
        engine = None
        if shard.ctx:
            shard.ctx.__dict__.pop("_floor_keep", None)
        import gc

        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        shard.close()
