"""ctypes binding of the libhydrium C API (include/libhydrium/libhydrium.h).

The binding is deliberately library-agnostic: ``Library(path)`` binds the nine ``hyd_*`` entry
points of *any* shared object exporting them, so the same driver code runs the MI355X build
(``hydrium_amd/lib/libhydrium.so.0``) and — in tests only — a build of the reference
itself.  ``encode_image`` reproduces the reference CLI's call pattern
(reference src/hydrium.c:275-286, 402-479): one-frame mode by default, tiles row-major, a
fixed-size output buffer cycled through flush / release / provide.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

HYD_OK = 0
HYD_NEED_MORE_OUTPUT = -2
HYD_ERROR_START = -10
HYD_NOMEM = -13
HYD_API_ERROR = -14
HYD_INTERNAL_ERROR = -15

HYD_UINT8, HYD_UINT16, HYD_FLOAT32 = 0, 1, 2

_HERE = os.path.dirname(os.path.abspath(__file__))
# HYDAMD_LIB: another build of the same library (probe builds of scripts/probe_k1_phases.py); never a fallback
DEFAULT_LIB = os.environ.get("HYDAMD_LIB") or os.path.join(_HERE, "lib", "libhydrium.so.0")


class HYDImageMetadata(C.Structure):
    _fields_ = [
        ("width", C.c_size_t),
        ("height", C.c_size_t),
        ("linear_light", C.c_int),
        ("tile_size_shift_x", C.c_int),
        ("tile_size_shift_y", C.c_int),
    ]


class HydriumError(RuntimeError):
    def __init__(self, code: int, message: Optional[str]):
        super().__init__(f"libhydrium error {code}: {message}")
        self.code = code
        self.message = message


class Library:
    """The nine exported functions of a libhydrium-compatible shared object."""

    SYMBOLS = (
        "hyd_encoder_new", "hyd_encoder_destroy", "hyd_set_metadata", "hyd_provide_output_buffer",
        "hyd_release_output_buffer", "hyd_flush", "hyd_send_tile", "hyd_error_message_get",
        "hyd_set_suggested_icc_profile",
    )

    def __init__(self, path: str = DEFAULT_LIB):
        if not os.path.exists(path):
            raise FileNotFoundError(
                f"{path} not found - build it first (python -c 'import __graft_entry__ as g; g.build()')")
        self.path = path
        from . import preload_hip_runtime

        preload_hip_runtime()
        self.dll = C.CDLL(path)
        d = self.dll
        d.hyd_encoder_new.restype = C.c_void_p
        d.hyd_encoder_new.argtypes = []
        d.hyd_encoder_destroy.restype = C.c_int
        d.hyd_encoder_destroy.argtypes = [C.c_void_p]
        d.hyd_set_metadata.restype = C.c_int
        d.hyd_set_metadata.argtypes = [C.c_void_p, C.POINTER(HYDImageMetadata)]
        d.hyd_provide_output_buffer.restype = C.c_int
        d.hyd_provide_output_buffer.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        d.hyd_release_output_buffer.restype = C.c_int
        d.hyd_release_output_buffer.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
        d.hyd_flush.restype = C.c_int
        d.hyd_flush.argtypes = [C.c_void_p]
        d.hyd_send_tile.restype = C.c_int
        d.hyd_send_tile.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_uint32, C.c_uint32,
                                    C.c_ssize_t, C.c_ssize_t, C.c_int, C.c_int]
        d.hyd_error_message_get.restype = C.c_char_p
        d.hyd_error_message_get.argtypes = [C.c_void_p]
        d.hyd_set_suggested_icc_profile.restype = C.c_int
        d.hyd_set_suggested_icc_profile.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        # additive knob of this build (include/hydrium_amd.h); the reference library has no such symbol
        self.has_tile_pipeline = hasattr(d, "hydamd_set_tile_pipeline")
        if self.has_tile_pipeline:
            d.hydamd_set_tile_pipeline.restype = C.c_int
            d.hydamd_set_tile_pipeline.argtypes = [C.c_void_p, C.c_int]
            d.hydamd_get_tile_pipeline.restype = C.c_int
            d.hydamd_get_tile_pipeline.argtypes = [C.c_void_p]


_FMT = {np.dtype(np.uint8): HYD_UINT8, np.dtype(np.uint16): HYD_UINT16, np.dtype(np.float32): HYD_FLOAT32}


class Encoder:
    """One HYDEncoder session.  Methods return the raw status code; ``check`` raises on errors."""

    def __init__(self, lib: Library):
        self.lib = lib
        self.h = lib.dll.hyd_encoder_new()
        if not self.h:
            raise MemoryError("hyd_encoder_new returned NULL")
        self._keep = None

    def close(self):
        if self.h:
            self.lib.dll.hyd_encoder_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def error_message(self) -> Optional[str]:
        m = self.lib.dll.hyd_error_message_get(self.h)
        return m.decode() if m else None

    def check(self, code: int) -> int:
        if code < HYD_ERROR_START:
            raise HydriumError(code, self.error_message())
        return code

    def set_metadata(self, width, height, linear_light=0, shift_x=-1, shift_y=-1) -> int:
        md = HYDImageMetadata(width, height, linear_light, shift_x, shift_y)
        return self.lib.dll.hyd_set_metadata(self.h, C.byref(md))

    def set_tile_pipeline(self, depth: int) -> int:
        """Tile-mode frames in flight (hydamd_set_tile_pipeline): 1 = the reference's timing, up to 8; 0 = process default."""
        return self.lib.dll.hydamd_set_tile_pipeline(self.h, depth)

    def tile_pipeline(self) -> int:
        return self.lib.dll.hydamd_get_tile_pipeline(self.h)

    def set_icc(self, icc: Optional[bytes]) -> int:
        if icc is None:
            return self.lib.dll.hyd_set_suggested_icc_profile(self.h, None, 0)
        return self.lib.dll.hyd_set_suggested_icc_profile(self.h, icc, len(icc))

    def provide_output(self, buf: C.Array) -> int:
        self._keep = buf
        return self.lib.dll.hyd_provide_output_buffer(self.h, C.cast(buf, C.c_void_p), len(buf))

    def provide_output_raw(self, ptr, length) -> int:
        return self.lib.dll.hyd_provide_output_buffer(self.h, ptr, length)

    def release_output(self):
        n = C.c_size_t(0)
        code = self.lib.dll.hyd_release_output_buffer(self.h, C.byref(n))
        return code, n.value

    def flush(self) -> int:
        return self.lib.dll.hyd_flush(self.h)

    def send_tile_ptrs(self, ptrs, tile_x, tile_y, row_stride, pixel_stride, is_last, fmt) -> int:
        arr = (C.c_void_p * 3)(*ptrs)
        return self.lib.dll.hyd_send_tile(self.h, arr, tile_x, tile_y, row_stride, pixel_stride, is_last, fmt)

    def send_tile(self, img: np.ndarray, tile_x: int, tile_y: int, tile_w: int, tile_h: int,
                  is_last: int = -1, layout: str = "packed") -> int:
        """Send the tile at (tile_x, tile_y) of a full interleaved (H, W, 3) image.

        ``layout``: "packed" (pixel_stride 3 on the interleaved array), "planar" (three separate
        planes, pixel_stride 1) or "flipped" (packed, negative row stride, the PFM pattern of
        reference src/hydrium.c:456-460).
        """
        h, w, _ = img.shape
        fmt = _FMT[img.dtype]
        isz = img.dtype.itemsize
        x0, y0 = tile_x * tile_w, tile_y * tile_h
        if layout == "planar":
            planes = getattr(self, "_planes", None)
            if planes is None or planes[0] is not img:
                planes = (img, [np.ascontiguousarray(img[:, :, c]) for c in range(3)])
                self._planes = planes
            ptrs = [p.ctypes.data + (y0 * w + x0) * isz for p in planes[1]]
            return self.send_tile_ptrs(ptrs, tile_x, tile_y, w, 1, is_last, fmt)
        assert img.flags["C_CONTIGUOUS"]
        base = img.ctypes.data
        if layout == "packed":
            off = (y0 * w + x0) * 3 * isz
            ptrs = [base + off + c * isz for c in range(3)]
            return self.send_tile_ptrs(ptrs, tile_x, tile_y, 3 * w, 3, is_last, fmt)
        if layout == "flipped":
            # caller passes an image stored bottom-up; row y of the picture is storage row h-1-y
            off = ((h - 1 - y0) * w + x0) * 3 * isz
            ptrs = [base + off + c * isz for c in range(3)]
            return self.send_tile_ptrs(ptrs, tile_x, tile_y, -3 * w, 3, is_last, fmt)
        raise ValueError(layout)


def tile_dims(width: int, height: int, shift_x: int, shift_y: int):
    """Tile size in pixels as the library sees it (reference encoder.c:441-446)."""
    if shift_x < 0 or shift_y < 0:
        return 2048, 2048
    return 256 << shift_x, 256 << shift_y


def encode_image(lib: Library, img: np.ndarray, *, linear_light: int = 0, shift_x: int = -1, shift_y: int = -1,
                 out_buf_size: int = 1 << 20, layout: str = "packed", order=None, icc: Optional[bytes] = None,
                 explicit_last: bool = False, out_buf=None, tile_pipeline: Optional[int] = None, in_place: bool = False):
    """Encode a whole (H, W, 3) image the way the reference CLI does; returns the codestream.
    ``tile_pipeline``: tile-mode frames in flight (this build only; None leaves the default, one frame per call).
    ``in_place``: with an ``out_buf`` that took the whole file in one piece, return a memoryview of it instead of a bytes
    copy — where a C caller finds its file when hyd_release_output_buffer has returned (a 12 MB bytes object costs
    Python a millisecond)."""
    h, w, _ = img.shape
    src = img[::-1].copy() if layout == "flipped" else img
    tw, th = tile_dims(w, h, shift_x, shift_y)
    ntx, nty = -(-w // tw), -(-h // th)
    tiles = [(tx, ty) for ty in range(nty) for tx in range(ntx)] if order is None else list(order)
    chunks = []
    with Encoder(lib) as enc:
        enc.check(enc.set_metadata(w, h, linear_light, shift_x, shift_y))
        if tile_pipeline is not None:
            enc.check(enc.set_tile_pipeline(tile_pipeline))
        if icc is not None:
            enc.check(enc.set_icc(icc))
        buf = out_buf if out_buf is not None else (C.c_uint8 * out_buf_size)()  # a caller encoding many images keeps one buffer
        enc.check(enc.provide_output(buf))
        for i, (tx, ty) in enumerate(tiles):
            is_last = -1
            if explicit_last or order is not None:
                is_last = 1 if i == len(tiles) - 1 else 0
            ret = enc.check(enc.send_tile(src, tx, ty, tw, th, is_last, layout))
            if in_place and out_buf is not None and (shift_x < 0 or shift_y < 0) and i != len(tiles) - 1:
                continue  # one-frame mode: nothing but the file header exists before the final tile; it waits in `buf`
            while True:
                ret = enc.check(enc.flush())
                code, n = enc.release_output()
                enc.check(code)
                if n:
                    whole = in_place and out_buf is not None and not chunks and ret != HYD_NEED_MORE_OUTPUT and i == len(tiles) - 1
                    chunks.append(n if whole else C.string_at(buf, n))  # (whole: nothing else will be written to buf)
                enc.check(enc.provide_output(buf))
                if ret != HYD_NEED_MORE_OUTPUT:
                    break
    if len(chunks) == 1 and isinstance(chunks[0], int):
        return memoryview(buf).cast("B")[:chunks[0]]
    return chunks[0] if len(chunks) == 1 else b"".join(chunks)
