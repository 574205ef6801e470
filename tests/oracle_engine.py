"""A shard engine for hydrium_amd.multigpu.choreograph_frame that needs no GPU: the CPU oracle codes
the shard's LF groups, tests/lf_model.py their LF streams, and the results are packed into the blob
layout hydamd_export_frame produces (include/hydrium_amd.h HydAmdBlobHeader / HydAmdBlobSlot).

TEST INFRASTRUCTURE ONLY: lets the multi-process choreography — partition, all-gather of alphabet
maxima, floor per rank, one gather of blobs, hydamd_frame_from_blobs — run under gloo on CPU."""
from __future__ import annotations

import numpy as np
import torch

from hydrium_amd import device
from oracle import binding as orc

import lf_model


class OracleShardEngine:
    def __init__(self, img: np.ndarray, lf_ids, fail_first_export: bool = False):
        self.img, self.lf_ids = img, list(lf_ids)
        h, w, _ = img.shape
        self.lfx = -(-w // 2048)
        self.n_lf = self.lfx * (-(-h // 2048))
        self.results = None
        self.floor = None
        self.exports = 0
        self.fail_first_export = fail_first_export  # act like a shard whose frame outgrew its buffers once

    def _code(self, lf, running):
        return orc.encode_lf_group(self.img, lf % self.lfx, lf // self.lfx, num_presets=self.n_lf, preset=lf,
                                   max_alphabet_size=running)

    def enqueue_transform(self):
        # an LF group's own largest token + 1 does not depend on what ran before it
        self.maxima = [int(self._code(lf, 0)[0].alphabet_size.max()) for lf in self.lf_ids]

    def alphabet_maxima(self):
        return torch.tensor(self.maxima, dtype=torch.int32)

    def enqueue_entropy(self, floor_tensor):
        running = int(floor_tensor.item()) if floor_tensor is not None else 0
        self.results = []
        for lf in self.lf_ids:
            r, running = self._code(lf, running)
            self.results.append((lf, r, running))

    def blob_bound(self) -> int:
        return 64 + len(self.lf_ids) * device.BLOB_SLOT_DTYPE.itemsize + (1 << 20)

    def export_blob(self, out):
        self.exports += 1
        slots = np.zeros(len(self.results), device.BLOB_SLOT_DTYPE)
        lf_bytes, hf = b"", b""
        for s, (lf, r, running) in enumerate(self.results):
            _, lengths, alphabet, pairs, packed, nbits = lf_model.model(r.dc)
            ncl = r.cluster_to - r.cluster_from
            slots[s]["preset"] = lf
            slots[s]["running_max_alphabet"] = running
            slots[s]["log_alphabet_size"] = r.log_alphabet_size
            slots[s]["alphabet"][:ncl] = r.alphabet_size[r.cluster_from:r.cluster_to]
            slots[s]["group_bits"][:r.num_groups] = r.group_bits
            slots[s]["freq"][:ncl] = r.freqs[r.cluster_from:r.cluster_to]
            slots[s]["lf"]["bit_count"], slots[s]["lf"]["alphabet"], slots[s]["lf"]["run_pairs"] = nbits, alphabet, pairs
            slots[s]["lf"]["offset"] = len(lf_bytes)
            slots[s]["lf"]["lengths"] = np.asarray(lengths, np.uint8)
            lf_bytes += bytes(np.asarray(packed, np.uint8)) + b"\0" * (-len(packed) % 4)
            hf += r.stream
        head = np.zeros(1, device.BLOB_HEADER_DTYPE)
        lf_off = 64 + slots.nbytes
        hf_off = (lf_off + len(lf_bytes) + 15) & ~15
        head["magic"], head["version"], head["num_slots"], head["lf_coded"] = device.BLOB_MAGIC, 1, len(slots), 1
        head["hf_bytes"], head["lf_bytes"], head["total_bytes"] = len(hf), len(lf_bytes), hf_off + len(hf)
        if self.fail_first_export and self.exports == 1:
            head["status"] = 2  # HYDK_STATUS_TOKENS: "this shard has to run the frame again"
        blob = head.tobytes() + slots.tobytes() + lf_bytes
        blob += b"\0" * (hf_off - len(blob)) + hf
        assert len(blob) <= out.numel()
        out[:len(blob)] = torch.from_numpy(np.frombuffer(blob, np.uint8).copy())

    def finish(self):
        pass
