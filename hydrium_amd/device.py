"""ctypes binding of the additive device-level C-ABI (include/hydrium_amd.h).

This is plumbing for tests, ``bench.py`` and the multi-GPU driver: it hands raw device pointers
(from torch tensors) to the HIP hot path and reads results back.  There is no CPU fallback — if the
library or a GPU is missing, construction fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

from . import api

MAX_CLUSTERS = 9
ALPHABET = 128
GROUPS_PER_LFG = 64
K_NAMES = ("transform_tokenize", "build_tables", "rans_encode", "pack_sections", "lf_coder")
LF_INFO_DTYPE = np.dtype([("bit_count", "<u4"), ("alphabet", "<u4"), ("run_pairs", "<u4"), ("error", "<u4"),
                          ("offset", "<u4"), ("reserved", "<u4", (3,)), ("lengths", "u1", (384,))])
LF_BITWORDS = 3 * 256 * 256 * 2 + 2                     # HYDK_LF_BITWORDS
LF_CODES = 384  # compact token space of the LF-coefficient stream (include/hydrium_amd.h HYDAMD_LF_CODES)

FMT_OF_DTYPE = {np.dtype(np.uint8): 0, np.dtype(np.uint16): 1, np.dtype(np.float32): 2}


class DeviceError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"hydamd error {code}: {msg}")
        self.code = code
        self.message = msg


_dll = None


def dll(path: Optional[str] = None):
    global _dll
    if _dll is None or path is not None:
        p = path or api.DEFAULT_LIB
        if not os.path.exists(p):
            raise FileNotFoundError(f"{p} missing: the HIP extension is not built (there is no CPU fallback)")
        from . import preload_hip_runtime

        preload_hip_runtime()
        d = C.CDLL(p)
        vp, i, u, sz = C.c_void_p, C.c_int, C.c_uint, C.c_size_t
        d.hydamd_device_count.restype = i
        d.hydamd_create.restype = vp
        d.hydamd_create.argtypes = [i, i, i, i, C.POINTER(i)]
        d.hydamd_destroy.argtypes = [vp]
        d.hydamd_error.restype = C.c_char_p
        d.hydamd_error.argtypes = [vp]
        d.hydamd_set_stream.argtypes = [vp, vp]
        d.hydamd_get_stream.restype = vp
        d.hydamd_get_stream.argtypes = [vp]
        d.hydamd_uses_register_luts.argtypes = [vp]
        d.hydamd_force_luts.argtypes = [vp, i]
        d.hydamd_xyb_mode.argtypes = [vp]
        d.hydamd_set_xyb_mode.argtypes = [vp, i]
        d.hydamd_begin_frame.argtypes = [vp, u]
        d.hydamd_set_rans_waves.argtypes = [vp, i]
        d.hydamd_set_curve_gathers.argtypes = [vp, i]
        lf_args = [vp, i, C.POINTER(vp), C.c_ssize_t, C.c_ssize_t, i, sz, sz, u]
        d.hydamd_encode_lf_group.argtypes = lf_args
        d.hydamd_encode_lf_group_host.argtypes = lf_args
        d.hydamd_encode_image.argtypes = [vp, C.POINTER(vp), C.c_ssize_t, C.c_ssize_t, i, sz, sz]
        d.hydamd_debug_shader_clock_mhz.argtypes = [vp, C.POINTER(C.c_double)]
        d.hydamd_encode_image_batch.argtypes = [vp, i, C.POINTER(vp), C.c_ssize_t, C.c_ssize_t, i, sz, sz]
        d.hydamd_begin_batch.argtypes = [vp, u, i]
        d.hydamd_finish_frame.argtypes = [vp, i]
        d.hydamd_run_transform.argtypes = [vp, i]
        d.hydamd_run_entropy.argtypes = [vp, i]
        d.hydamd_read_alphabet_max.argtypes = [vp, i, C.POINTER(C.c_uint32)]
        d.hydamd_set_alphabet_floor.argtypes = [vp, C.c_uint32]
        d.hydamd_alphabet_max_device.restype = vp
        d.hydamd_alphabet_max_device.argtypes = [vp]
        d.hydamd_set_alphabet_floor_device.argtypes = [vp, vp]
        d.hydamd_blob_bound.restype = sz
        d.hydamd_blob_bound.argtypes = [vp, i]
        d.hydamd_export_frame.argtypes = [vp, i, vp, sz]
        d.hydamd_export_frame_owned.argtypes = [vp, i, C.POINTER(vp), C.POINTER(sz)]
        d.hydamd_frame_from_blobs.restype = i
        d.hydamd_frame_from_blobs.argtypes = [C.POINTER(api.HYDImageMetadata), i, i, sz, C.POINTER(vp), C.POINTER(sz),
                                              C.c_char_p, sz, C.POINTER(vp), C.POINTER(sz), C.POINTER(C.c_char_p)]
        d.hydamd_frame_from_results.restype = i
        d.hydamd_frame_from_results.argtypes = [
            C.POINTER(api.HYDImageMetadata), i, i, sz, vp, C.POINTER(vp), vp, vp, vp, u, vp, sz, C.c_char_p, sz,
            C.POINTER(vp), C.POINTER(sz), C.POINTER(C.c_char_p)]
        d.hydamd_frame_from_streams.restype = i
        d.hydamd_frame_from_streams.argtypes = [
            C.POINTER(api.HYDImageMetadata), i, i, sz, vp, vp, vp, vp, vp, u, C.c_char_p, sz, C.c_char_p, sz,
            C.POINTER(vp), C.POINTER(sz), C.POINTER(C.c_char_p)]
        d.hydamd_free.argtypes = [vp]
        d.hydamd_sync.argtypes = [vp]
        d.hydamd_payload_size.restype = sz
        d.hydamd_payload_size.argtypes = [vp]
        d.hydamd_payload_capacity.restype = sz
        d.hydamd_payload_capacity.argtypes = [vp]
        d.hydamd_token_capacity.restype = u
        d.hydamd_token_capacity.argtypes = [vp]
        d.hydamd_overflow_reruns.restype = u
        d.hydamd_overflow_reruns.argtypes = [vp]
        d.hydamd_grown_ahead.restype = u
        d.hydamd_grown_ahead.argtypes = [vp]
        d.hydamd_payload_device.restype = vp
        d.hydamd_payload_device.argtypes = [vp]
        d.hydamd_read_payload.argtypes = [vp, vp, sz]
        d.hydamd_read_sections.argtypes = [vp, i, vp, vp]
        d.hydamd_read_tables.argtypes = [vp, i, vp, vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        d.hydamd_read_dc.argtypes = [vp, i, vp, sz, sz]
        d.hydamd_read_symbol_counts.argtypes = [vp, i, vp]
        d.hydamd_read_tokens.argtypes = [vp, i, i, vp, sz]
        d.hydamd_read_debug_plane.argtypes = [vp, i, vp, sz, sz]
        d.hydamd_submit_lf_group.argtypes = [vp, i]
        d.hydamd_run_lf_coder.argtypes = [vp, i, i]
        d.hydamd_sync_lf.argtypes = [vp]
        d.hydamd_set_lf_coder.argtypes = [vp, i]
        d.hydamd_lf_coder.argtypes = [vp]
        d.hydamd_read_lf_stream.argtypes = [vp, i, vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        d.hydamd_read_lf_bits.argtypes = [vp, i, vp, sz]
        d.hydamd_read_lf_streams.argtypes = [vp, i, i, vp]
        d.hydamd_lf_payload_size.restype = sz
        d.hydamd_lf_payload_size.argtypes = [vp]
        d.hydamd_lf_payload_device.restype = vp
        d.hydamd_lf_payload_device.argtypes = [vp]
        d.hydamd_read_lf_payload.argtypes = [vp, vp, sz]
        d.hydamd_debug_transform_footprint.restype = C.c_int
        d.hydamd_debug_transform_footprint.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        d.hydamd_debug_lf_code.argtypes = [vp, vp, vp, vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        d.hydamd_assembler_create.restype = vp
        d.hydamd_assembler_create.argtypes = [i, C.POINTER(i)]
        d.hydamd_assembler_destroy.argtypes = [vp]
        d.hydamd_assembler_error.restype = C.c_char_p
        d.hydamd_assembler_error.argtypes = [vp]
        d.hydamd_assembler_plan.argtypes = [vp, C.POINTER(api.HYDImageMetadata), i, i, sz, vp, vp, C.c_char_p, sz]
        d.hydamd_assembler_run.argtypes = [vp, C.POINTER(vp), C.POINTER(sz), vp, vp, sz]
        d.hydamd_assembler_result.argtypes = [vp, C.POINTER(sz)]
        d.hydamd_profile.argtypes = [vp, i]
        d.hydamd_profile_read.argtypes = [vp, vp, vp]
        d.hydamd_multi_create.restype = vp
        d.hydamd_multi_create.argtypes = [i, C.POINTER(i), C.POINTER(api.HYDImageMetadata), C.POINTER(i)]
        d.hydamd_multi_destroy.argtypes = [vp]
        d.hydamd_multi_error.restype = C.c_char_p
        d.hydamd_multi_error.argtypes = [vp]
        d.hydamd_multi_context.restype = vp
        d.hydamd_multi_context.argtypes = [vp, i]
        d.hydamd_multi_shard_lf_groups.argtypes = [vp, i, C.POINTER(sz), C.POINTER(sz)]
        d.hydamd_encode_image_multi.argtypes = [vp, C.POINTER(vp), C.c_ssize_t, C.c_ssize_t, i, i]
        d.hydamd_multi_result.argtypes = [vp, C.POINTER(sz)]
        d.hydamd_multi_read.argtypes = [vp, vp, sz]
        if path is not None:
            return d
        _dll = d
    return _dll


class DeviceContext:
    """One HydAmdContext: one GPU, one stream, ``max_lf_groups`` LF-group slots."""

    def __init__(self, device: int = 0, max_lf_groups: int = 1, linear_light: int = 0, debug_planes: bool = False):
        self.d = dll()
        st = C.c_int(0)
        self.h = self.d.hydamd_create(device, max_lf_groups, linear_light, int(debug_planes), C.byref(st))
        if not self.h:
            raise DeviceError(st.value, (self.d.hydamd_error(None) or b"").decode())
        self.max_lf_groups = max_lf_groups

    def close(self):
        if self.h:
            self.d.hydamd_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _ck(self, code: int) -> int:
        if code != 0:
            raise DeviceError(code, (self.d.hydamd_error(self.h) or b"").decode())
        return code

    # -- control ---------------------------------------------------------------------------------
    def set_stream(self, stream_ptr: Optional[int]):
        self._ck(self.d.hydamd_set_stream(self.h, stream_ptr))

    def get_stream(self) -> int:
        return self.d.hydamd_get_stream(self.h) or 0

    def uses_register_luts(self) -> bool:
        return bool(self.d.hydamd_uses_register_luts(self.h))

    def xyb_mode(self) -> int:
        return self.d.hydamd_xyb_mode(self.h)

    def set_xyb_mode(self, mode: int):
        self._ck(self.d.hydamd_set_xyb_mode(self.h, mode))

    def force_luts(self, use_luts: bool):
        self._ck(self.d.hydamd_force_luts(self.h, int(use_luts)))

    def set_rans_waves(self, waves: int):
        self._ck(self.d.hydamd_set_rans_waves(self.h, waves))

    def set_curve_gathers(self, mode: int):
        """0: one curve of a pixel's six gathered unless the last frame was dense (noise); 1 always; 2 never"""
        self._ck(self.d.hydamd_set_curve_gathers(self.h, mode))

    def begin_frame(self, num_presets: int):
        self._ck(self.d.hydamd_begin_frame(self.h, num_presets))

    def encode_lf_group(self, slot: int, ptrs: Sequence[int], row_stride: int, pixel_stride: int, fmt: int,
                        width: int, height: int, preset: int, host: bool = False):
        arr = (C.c_void_p * 3)(*ptrs)
        fn = self.d.hydamd_encode_lf_group_host if host else self.d.hydamd_encode_lf_group
        self._ck(fn(self.h, slot, arr, row_stride, pixel_stride, fmt, width, height, preset))

    def finish_frame(self, num_slots: int):
        self._ck(self.d.hydamd_finish_frame(self.h, num_slots))

    def run_transform(self, num_slots: int):
        self._ck(self.d.hydamd_run_transform(self.h, num_slots))

    def run_entropy(self, num_slots: int):
        self._ck(self.d.hydamd_run_entropy(self.h, num_slots))

    def read_alphabet_max(self, slot: int) -> int:
        v = C.c_uint32(0)
        self._ck(self.d.hydamd_read_alphabet_max(self.h, slot, C.byref(v)))
        return v.value

    def set_alphabet_floor(self, floor: int):
        self._ck(self.d.hydamd_set_alphabet_floor(self.h, floor))

    # -- the same exchange and the result hand-over without the host in the loop (multi-GPU) ----------
    def alphabet_max_tensor(self, num_slots: int):
        """int32 CUDA view of the per-slot maxima in the context's memory; order reads behind the context's stream."""
        raw = self._device_view("alpha_max", int(self.d.hydamd_alphabet_max_device(self.h) or 0), self.max_lf_groups * 4)
        return raw.view(__import__("torch").int32)[:num_slots]

    def set_alphabet_floor_device(self, tensor):
        """1-element int32 CUDA tensor the table kernel reads the floor from when it runs (kept alive by the context object)."""
        self._floor_keep = tensor
        self._ck(self.d.hydamd_set_alphabet_floor_device(self.h, tensor.data_ptr() if tensor is not None else None))

    def blob_bound(self, num_slots: int) -> int:
        return int(self.d.hydamd_blob_bound(self.h, num_slots))

    def export_frame(self, num_slots: int, out_tensor):
        """Enqueue the blob of slots [0, num_slots) into a uint8 CUDA tensor (behind the entropy stage, no sync)."""
        self._ck(self.d.hydamd_export_frame(self.h, num_slots, out_tensor.data_ptr(), out_tensor.numel()))

    def export_frame_owned(self, num_slots: int):
        """Enqueue the blob of slots [0, num_slots) as a VIEW in the context's own small buffer (nothing of the frame's bulk
        is copied; for an assembler on the same device and stream).  Returns (device pointer, readable bytes)."""
        p, n = C.c_void_p(0), C.c_size_t(0)
        self._ck(self.d.hydamd_export_frame_owned(self.h, num_slots, C.byref(p), C.byref(n)))
        return int(p.value), int(n.value)

    def sync(self):
        self._ck(self.d.hydamd_sync(self.h))

    # -- whole-image helpers -----------------------------------------------------------------------
    def encode_image_tensor(self, img, num_slots_check: bool = True):
        """Enqueue the hot path for every LF group of an interleaved (H, W, 3) torch CUDA tensor.

        The kernels read the tensor's memory asynchronously: the caller must keep it alive (and
        unmodified) until ``sync()``."""
        h, w, _ = img.shape
        isz = img.element_size()
        fmt = {1: 0, 2: 1, 4: 2}[isz]
        lfx, lfy = -(-w // 2048), -(-h // 2048)
        n = lfx * lfy
        if num_slots_check and n > self.max_lf_groups:
            raise ValueError("context has too few LF-group slots for this image")
        if not getattr(self, "per_lf_group_calls", False):
            # one C call per frame (hydamd_encode_image): eighteen ctypes calls cost the host as much as the GPU needs
            base = img.data_ptr()
            arr = (C.c_void_p * 3)(base, base + isz, base + 2 * isz)
            self._ck(self.d.hydamd_encode_image(self.h, arr, 3 * w, 3, fmt, w, h))
            return n
        self.begin_frame(n)
        base = img.data_ptr()
        for ty in range(lfy):
            for tx in range(lfx):
                x0, y0 = tx * 2048, ty * 2048
                p = base + (y0 * w + x0) * 3 * isz
                self.encode_lf_group(ty * lfx + tx, [p, p + isz, p + 2 * isz], 3 * w, 3, fmt,
                                     min(2048, w - x0), min(2048, h - y0), ty * lfx + tx)
        self.finish_frame(n)
        return n

    def encode_image_batch(self, imgs):
        """Enqueue a batch of interleaved (H, W, 3) torch CUDA tensors of one shape as one launch group
        (hydamd_encode_image_batch): frame k in slots k * n ... (k + 1) * n - 1."""
        h, w, _ = imgs[0].shape
        isz = imgs[0].element_size()
        fmt = {1: 0, 2: 1, 4: 2}[isz]
        n = (-(-w // 2048)) * (-(-h // 2048))
        if n * len(imgs) > self.max_lf_groups:
            raise ValueError("context has too few LF-group slots for this batch")
        ptrs = []
        for t in imgs:
            assert t.shape == imgs[0].shape and t.element_size() == isz
            b = t.data_ptr()
            ptrs += [b, b + isz, b + 2 * isz]
        arr = (C.c_void_p * len(ptrs))(*ptrs)
        self._ck(self.d.hydamd_encode_image_batch(self.h, len(imgs), arr, 3 * w, 3, fmt, w, h))
        return n * len(imgs)

    def encode_image_host(self, img: np.ndarray):
        """Same from a host numpy image, through the pinned staging path."""
        h, w, _ = img.shape
        isz = img.dtype.itemsize
        fmt = FMT_OF_DTYPE[img.dtype]
        lfx, lfy = -(-w // 2048), -(-h // 2048)
        n = lfx * lfy
        self.begin_frame(n)
        base = img.ctypes.data
        for ty in range(lfy):
            for tx in range(lfx):
                x0, y0 = tx * 2048, ty * 2048
                p = base + (y0 * w + x0) * 3 * isz
                self.encode_lf_group(ty * lfx + tx, [p, p + isz, p + 2 * isz], 3 * w, 3, fmt,
                                     min(2048, w - x0), min(2048, h - y0), ty * lfx + tx, host=True)
        self.finish_frame(n)
        return n

    # -- results -----------------------------------------------------------------------------------
    def payload_size(self) -> int:
        return self.d.hydamd_payload_size(self.h)

    def payload_device_ptr(self) -> int:
        return self.d.hydamd_payload_device(self.h) or 0

    def _device_view(self, key: str, ptr: int, capacity: int):
        """uint8 CUDA tensor aliasing `capacity` bytes at `ptr`, made once per buffer: wrapping a raw
        pointer costs torch a pointer-attribute query (~0.5 ms), slicing the cached view nothing."""
        import torch

        cache = self.__dict__.setdefault("_views", {})
        hit = cache.get(key)
        if hit is None or hit[0] != ptr:
            class _View:
                __cuda_array_interface__ = {"shape": (capacity,), "typestr": "|u1", "data": (ptr, False), "version": 2}

            hit = cache[key] = (ptr, torch.as_tensor(_View(), device="cuda"))
        return hit[1]

    def payload_tensor(self):
        """Zero-copy torch uint8 view of the packed HF sections in HBM (valid until the next frame)."""
        cap = int(self.d.hydamd_payload_capacity(self.h))  # pointer and capacity change if a frame outgrew the buffers
        return self._device_view(("payload", cap), self.payload_device_ptr(), cap)[: self.payload_size()]

    def token_capacity(self) -> int:
        return int(self.d.hydamd_token_capacity(self.h))

    def grown_ahead(self) -> int:
        return int(self.d.hydamd_grown_ahead(self.h))

    def overflow_reruns(self) -> int:
        return int(self.d.hydamd_overflow_reruns(self.h))

    def read_payload(self) -> bytes:
        n = self.payload_size()
        buf = np.zeros(max(n, 1), np.uint8)
        self._ck(self.d.hydamd_read_payload(self.h, buf.ctypes.data, n))
        return buf[:n].tobytes()

    def read_sections(self, slot: int):
        bits = np.zeros(GROUPS_PER_LFG, np.uint32)
        offs = np.zeros(GROUPS_PER_LFG, np.uint64)
        self._ck(self.d.hydamd_read_sections(self.h, slot, bits.ctypes.data, offs.ctypes.data))
        return bits.astype(np.int64), offs.astype(np.int64)

    def read_tables(self, slot: int):
        freq = np.zeros((MAX_CLUSTERS, ALPHABET), np.uint32)
        alpha = np.zeros(MAX_CLUSTERS, np.uint32)
        la, rm = C.c_uint32(0), C.c_uint32(0)
        self._ck(self.d.hydamd_read_tables(self.h, slot, freq.ctypes.data, alpha.ctypes.data, C.byref(la), C.byref(rm)))
        return freq, alpha.astype(np.int64), la.value, rm.value

    def read_dc(self, slot: int, vbw: int, vbh: int) -> np.ndarray:
        out = np.zeros((3, vbh, vbw), np.int32)
        self._ck(self.d.hydamd_read_dc(self.h, slot, out.ctypes.data, vbw, vbh))
        return out

    def read_symbol_counts(self, slot: int) -> np.ndarray:
        out = np.zeros(GROUPS_PER_LFG, np.uint32)
        self._ck(self.d.hydamd_read_symbol_counts(self.h, slot, out.ctypes.data))
        return out.astype(np.int64)

    def read_tokens(self, slot: int, group: int, count: int) -> np.ndarray:
        out = np.zeros(max(count, 1), np.uint64)
        self._ck(self.d.hydamd_read_tokens(self.h, slot, group, out.ctypes.data, count))
        return out[:count]

    def read_debug_plane(self, which: int, pitch: int, rows: int) -> np.ndarray:
        out = np.zeros((3, rows, pitch), np.int32 if which == 2 else np.float32)
        self._ck(self.d.hydamd_read_debug_plane(self.h, which, out.ctypes.data, pitch, rows))
        return out

    def profile(self, enable: bool):
        self._ck(self.d.hydamd_profile(self.h, int(enable)))

    def profile_read(self):
        ms = (C.c_double * len(K_NAMES))()
        n = (C.c_uint64 * len(K_NAMES))()
        self._ck(self.d.hydamd_profile_read(self.h, ms, n))
        return {K_NAMES[i]: (ms[i], n[i]) for i in range(len(K_NAMES))}

    def submit_lf_group(self, slot: int):
        """Enqueue the transform stage of the LF group in `slot` now (slots in order) instead of at finish_frame."""
        self._ck(self.d.hydamd_submit_lf_group(self.h, slot))

    def run_lf_coder(self, num_slots: int, last: bool):
        """Enqueue the LF coder for the transformed slots below `num_slots` in the context's stream;
        `last` also packs the frame's LF streams so that sync_lf() can wait for just them."""
        self._ck(self.d.hydamd_run_lf_coder(self.h, num_slots, int(last)))

    def sync_lf(self):
        self._ck(self.d.hydamd_sync_lf(self.h))

    # ---- LF-group coder (csrc/hip/lf_coder.hip) ----
    def set_lf_coder(self, on_device):
        """False/0 off, True/1 on a side stream (lowest latency), 2 at the end of the main stream (throughput)."""
        self._ck(self.d.hydamd_set_lf_coder(self.h, int(on_device)))

    def lf_coder(self) -> bool:
        return bool(self.d.hydamd_lf_coder(self.h))

    def read_lf_stream(self, slot: int):
        """(lengths[384] u8, alphabet, run_pairs, bit_count) of the device-coded LF-coefficient stream."""
        lengths = np.zeros(LF_CODES, np.uint8)
        a, r, b = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        self._ck(self.d.hydamd_read_lf_stream(self.h, slot, lengths.ctypes.data, C.byref(a), C.byref(r), C.byref(b)))
        return lengths, a.value, r.value, b.value

    def read_lf_bits(self, slot: int, bit_count: int) -> np.ndarray:
        out = np.zeros(max((bit_count + 7) // 8, 1), np.uint8)
        self._ck(self.d.hydamd_read_lf_bits(self.h, slot, out.ctypes.data, (bit_count + 7) // 8))
        return out[: (bit_count + 7) // 8]

    def read_lf_streams(self, count: int, first: int = 0) -> np.ndarray:
        """Structured array of the per-slot LF stream records (bit_count, alphabet, run_pairs, error, offset, lengths)."""
        out = np.zeros(count, LF_INFO_DTYPE)
        self._ck(self.d.hydamd_read_lf_streams(self.h, first, count, out.ctypes.data))
        return out

    def lf_payload_size(self) -> int:
        return int(self.d.hydamd_lf_payload_size(self.h))

    def read_lf_payload(self) -> np.ndarray:
        n = self.lf_payload_size()
        out = np.zeros(max(n, 1), np.uint8)
        self._ck(self.d.hydamd_read_lf_payload(self.h, out.ctypes.data, n))
        return out[:n]

    def lf_payload_tensor(self):
        """The frame's packed LF symbol data as a CUDA uint8 tensor aliasing the context's buffer (valid after sync)."""
        import torch

        cap = self.max_lf_groups * LF_BITWORDS * 4
        return self._device_view("lf", int(self.d.hydamd_lf_payload_device(self.h) or 0), cap)[: self.lf_payload_size()]

    def shader_clock_mhz(self) -> float:
        v = C.c_double(0)
        self._ck(self.d.hydamd_debug_shader_clock_mhz(self.h, C.byref(v)))
        return v.value

    def transform_footprint(self, sample_fmt: int):
        """(static LDS bytes, registers per thread) of the transform kernel instance serving `sample_fmt` (0 u8, 1 u16, 2 f32)."""
        lds, regs = C.c_int(0), C.c_int(0)
        self._ck(self.d.hydamd_debug_transform_footprint(self.h, int(sample_fmt), C.byref(lds), C.byref(regs)))
        return lds.value, regs.value

    def debug_lf_code(self, hist: np.ndarray):
        """Device code construction for one histogram over the compact token space -> (lengths, codes, alphabet, error)."""
        hist = np.ascontiguousarray(hist, np.uint32)
        assert hist.shape == (LF_CODES,)
        lengths = np.zeros(LF_CODES, np.uint8)
        codes = np.zeros(LF_CODES, np.uint32)
        a, e = C.c_uint32(0), C.c_uint32(0)
        self._ck(self.d.hydamd_debug_lf_code(self.h, hist.ctypes.data, lengths.ctypes.data, codes.ctypes.data,
                                             C.byref(a), C.byref(e)))
        return lengths, codes, a.value, e.value


class HydAmdLfStream(C.Structure):
    _fields_ = [("lengths", C.c_void_p), ("alphabet", C.c_uint32), ("run_pairs", C.c_uint32), ("bits", C.c_void_p),
                ("bit_count", C.c_uint64)]


def frame_from_results(md: "api.HYDImageMetadata", tiles, dcs, freqs, alphabets, group_bits, max_alphabet: int,
                       payload: bytes, *, write_header=True, is_last=True, icc: Optional[bytes] = None,
                       lf_streams=None) -> bytes:
    """Codestream bytes from LF-group results in `tiles` order (host only).

    The LF coefficients come either as LF ints (``dcs``, hydamd_frame_from_results) or, when the GPU
    LF coder produced them, as ``lf_streams``: one (lengths, alphabet, run_pairs, bit_count, bits)
    tuple per LF group (hydamd_frame_from_streams)."""
    d = dll()
    n = len(tiles)
    tile_xy = np.ascontiguousarray(np.array(tiles, np.uint32).reshape(-1))
    freq = np.ascontiguousarray(np.stack(freqs).astype(np.uint32))
    alpha = np.ascontiguousarray(np.stack(alphabets).astype(np.uint32))
    bits = np.ascontiguousarray(np.stack(group_bits).astype(np.uint32))
    out, out_len, err = C.c_void_p(0), C.c_size_t(0), C.c_char_p(None)
    tail = (freq.ctypes.data, alpha.ctypes.data, bits.ctypes.data, max_alphabet, payload, len(payload), icc,
            len(icc) if icc else 0, C.byref(out), C.byref(out_len), C.byref(err))
    if lf_streams is not None:
        keep = [(np.ascontiguousarray(l, np.uint8), np.ascontiguousarray(b, np.uint8)) for l, _, _, _, b in lf_streams]
        arr = (HydAmdLfStream * n)()
        for i, (_, a, r, nb, _) in enumerate(lf_streams):
            arr[i] = HydAmdLfStream(keep[i][0].ctypes.data, a, r, keep[i][1].ctypes.data if nb else None, nb)
        ret = d.hydamd_frame_from_streams(C.byref(md), int(write_header), int(is_last), n, tile_xy.ctypes.data, arr, *tail)
    else:
        dc_arrays = [np.ascontiguousarray(a, np.int32) for a in dcs]
        dcp = (C.c_void_p * n)(*[a.ctypes.data for a in dc_arrays])
        ret = d.hydamd_frame_from_results(C.byref(md), int(write_header), int(is_last), n, tile_xy.ctypes.data, dcp, *tail)
    if ret:
        raise DeviceError(ret, (err.value or b"").decode())
    data = C.string_at(out.value, out_len.value)
    d.hydamd_free(out)
    return data


BLOB_HEADER_DTYPE = np.dtype([("magic", "<u4"), ("version", "<u4"), ("num_slots", "<u4"), ("status", "<u4"),
                              ("hf_bytes", "<u8"), ("lf_bytes", "<u8"), ("total_bytes", "<u8"), ("lf_coded", "<u4"),
                              ("reserved", "<u4", (5,))])
BLOB_SLOT_DTYPE = np.dtype([("preset", "<u4"), ("running_max_alphabet", "<u4"), ("log_alphabet_size", "<u4"),
                            ("table_error", "<u4"), ("alphabet", "<u4", (MAX_CLUSTERS,)), ("reserved", "<u4", (3,)),
                            ("group_bits", "<u4", (GROUPS_PER_LFG,)), ("freq", "<u4", (MAX_CLUSTERS, ALPHABET)),
                            ("lf", LF_INFO_DTYPE)])
BLOB_MAGIC = 0x42445948
BLOB_RETRY = 0xE


def blob_header(blob) -> np.ndarray:
    """The header record of a blob held in a bytes-like / uint8 array."""
    return np.frombuffer(memoryview(blob)[:BLOB_HEADER_DTYPE.itemsize], BLOB_HEADER_DTYPE)[0]


def frame_from_blobs(md: "api.HYDImageMetadata", blobs, *, write_header=True, is_last=True, icc: Optional[bytes] = None,
                     lib=None, raw: bool = False):
    """One-frame codestream from the blobs of the contexts that coded its LF groups (host only)."""
    d = lib or dll()
    arrs = [np.ascontiguousarray(np.frombuffer(b, np.uint8)) for b in blobs]
    ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    sizes = (C.c_size_t * len(arrs))(*[a.size for a in arrs])
    out, out_len, err = C.c_void_p(0), C.c_size_t(0), C.c_char_p(None)
    ret = d.hydamd_frame_from_blobs(C.byref(md), int(write_header), int(is_last), len(arrs), ptrs, sizes, icc,
                                    len(icc) if icc else 0, C.byref(out), C.byref(out_len), C.byref(err))
    if ret:
        raise DeviceError(ret, (err.value or b"").decode())
    if raw:  # the library's own buffer, no copy: (ctypes uint8 array, release())
        view = (C.c_uint8 * out_len.value).from_address(out.value)
        return view, (lambda: d.hydamd_free(out))
    data = C.string_at(out.value, out_len.value)
    d.hydamd_free(out)
    return data


class Assembler:
    """Device-side frame assembly (hydamd_assembler_*): shard blobs in device memory -> the finished one-frame
    codestream in one device-accessible buffer.  One assembler serves one frame at a time."""

    def __init__(self, device: int = 0):
        self.d = dll()
        st = C.c_int(0)
        self.h = self.d.hydamd_assembler_create(device, C.byref(st))
        if not self.h:
            raise DeviceError(st.value, "assembler could not be created")

    def close(self):
        if self.h:
            self.d.hydamd_assembler_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _ck(self, code: int):
        if code != 0:
            raise DeviceError(code, (self.d.hydamd_assembler_error(self.h) or b"").decode())

    def plan(self, md: "api.HYDImageMetadata", blob_lf_ids, *, write_header=True, is_last=True, icc: Optional[bytes] = None):
        """blob_lf_ids[b] = raster ids of the LF groups blob b carries, in its slot order (cheap when it repeats)."""
        counts = np.array([len(ids) for ids in blob_lf_ids], np.uint32)
        flat = np.array([lf for ids in blob_lf_ids for lf in ids], np.uint32)
        self._ck(self.d.hydamd_assembler_plan(self.h, C.byref(md), int(write_header), int(is_last), len(counts), counts.ctypes.data,
                                              flat.ctypes.data, icc, len(icc) if icc else 0))
        self.nblobs = len(counts)

    def run(self, blob_ptrs, blob_caps, out_ptr: int, out_cap: int, stream_ptr: int):
        """Enqueue on `stream_ptr` (a hipStream_t): blob_ptrs are device pointers, out_ptr any device-accessible buffer."""
        n = len(blob_ptrs)
        ptrs = (C.c_void_p * n)(*blob_ptrs)
        caps = (C.c_size_t * n)(*blob_caps)
        self._ck(self.d.hydamd_assembler_run(self.h, ptrs, caps, stream_ptr, out_ptr, out_cap))

    def run_tensors(self, blobs, out, stream=None):
        import torch

        st = stream if stream is not None else torch.cuda.current_stream()
        self.run([b.data_ptr() for b in blobs], [b.numel() for b in blobs], out.data_ptr(), out.numel(), st.cuda_stream)

    def result(self) -> int:
        """After the stream has been synchronised: bytes of the frame (raises on a device-side failure)."""
        n = C.c_size_t(0)
        self._ck(self.d.hydamd_assembler_result(self.h, C.byref(n)))
        return int(n.value)


class MultiFrame:
    """One device-resident frame on N devices of this process, composed in C (hydamd_multi_*, csrc/host/multi.c):
    LF groups dealt in raster runs, floors by peer read, the file assembled on a shard of the caller's choice."""

    def __init__(self, devices: Sequence[int], width: int, height: int, linear_light: int = 0):
        self.d = dll()
        md = api.HYDImageMetadata(width, height, int(linear_light), -1, -1)
        arr = (C.c_int * len(devices))(*devices)
        st = C.c_int(0)
        self.h = self.d.hydamd_multi_create(len(devices), arr, C.byref(md), C.byref(st))
        if not self.h:
            raise DeviceError(st.value, "multi-device frame could not be created")
        self.n, self.width, self.height = len(devices), width, height

    def close(self):
        if self.h:
            self.d.hydamd_multi_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _ck(self, code: int):
        if code != 0:
            raise DeviceError(code, (self.d.hydamd_multi_error(self.h) or b"").decode())

    def shard_lf_groups(self, shard: int):
        a, b = C.c_size_t(0), C.c_size_t(0)
        self._ck(self.d.hydamd_multi_shard_lf_groups(self.h, shard, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def encode(self, origins, assembling_shard: int = 0):
        """origins[d]: an interleaved (H, W, 3) torch tensor on shard d's device holding (at least) that shard's LF
        groups at their place in the image — or an int: the device address pixel (0, 0) would have.  Asynchronous; the
        tensors must stay alive until result()."""
        ptrs, isz = [], None
        for o in origins:
            if hasattr(o, "data_ptr"):
                isz = o.element_size()
                b = o.data_ptr()
            else:
                b = int(o)
            ptrs.append(b)
        isz = isz or getattr(self, "sample_bytes", 1)
        flat = []
        for b in ptrs:
            flat += [b, b + isz, b + 2 * isz]
        arr = (C.c_void_p * len(flat))(*flat)
        self._ck(self.d.hydamd_encode_image_multi(self.h, arr, 3 * self.width, 3, {1: 0, 2: 1, 4: 2}[isz], assembling_shard))

    def result(self) -> int:
        n = C.c_size_t(0)
        self._ck(self.d.hydamd_multi_result(self.h, C.byref(n)))
        return int(n.value)

    def read(self, out: Optional[np.ndarray] = None) -> np.ndarray:
        size = self.result()
        buf = out if out is not None and out.nbytes >= size else np.empty(size, np.uint8)
        self._ck(self.d.hydamd_multi_read(self.h, buf.ctypes.data, buf.nbytes))
        return buf[:size]

    def context_overflow_reruns(self, shard: int) -> int:
        return int(self.d.hydamd_overflow_reruns(self.d.hydamd_multi_context(self.h, shard)))


def decode_token_records(rec: np.ndarray):
    """Split device token records into (token, local cluster, residue_bits, residue) arrays."""
    lo = (rec & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    return lo & 0xFF, (lo >> 8) & 0xF, (lo >> 16) & 0x3F, (rec >> np.uint64(32)).astype(np.uint32)
