"""ctypes binding of oracle/liboracle.so (hyd_oracle.h).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")
MAX_ALPHABET = 128

FMT = {np.dtype(np.uint8): 0, np.dtype(np.uint16): 1, np.dtype(np.float32): 2}

SYMBOL_DTYPE = np.dtype([("token", "<u2"), ("cluster", "u1"), ("residue_bits", "u1"), ("residue", "<u4")])


class _Result(C.Structure):
    _fields_ = [
        ("width", C.c_size_t), ("height", C.c_size_t), ("vbw", C.c_size_t), ("vbh", C.c_size_t),
        ("stride", C.c_size_t), ("gcols", C.c_size_t), ("grows", C.c_size_t), ("num_groups", C.c_size_t),
        ("xyb", C.POINTER(C.c_float)), ("dct", C.POINTER(C.c_float)), ("quant", C.POINTER(C.c_int32)),
        ("dc", C.POINTER(C.c_int32)), ("nz", C.POINTER(C.c_uint8)),
        ("symbols", C.c_void_p), ("num_symbols", C.c_size_t), ("group_symbols", C.POINTER(C.c_size_t)),
        ("cluster_from", C.c_uint), ("cluster_to", C.c_uint),
        ("alphabet_size", C.c_uint16 * 256),
        ("freqs", (C.c_uint32 * MAX_ALPHABET) * 256),
        ("max_alphabet_size", C.c_uint), ("log_alphabet_size", C.c_int),
        ("stream", C.POINTER(C.c_uint8)), ("group_offset", C.POINTER(C.c_size_t)),
        ("group_bits", C.POINTER(C.c_size_t)), ("stream_bytes", C.c_size_t),
    ]


def build(force: bool = False) -> str:
    """Compile liboracle.so (and, when /root/reference exists, oracle/_ref) via the Makefile."""
    if force or not os.path.exists(LIB_PATH) or \
            os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "hyd_oracle.c")):
        subprocess.run(["make", "-C", _HERE, "liboracle.so"], check=True, capture_output=True)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        d = C.CDLL(LIB_PATH)
        d.orc_build_input_lut.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
        d.orc_build_bias_lut.argtypes = [C.c_void_p]
        d.orc_linearize.restype = C.c_float
        d.orc_linearize.argtypes = [C.c_float]
        d.orc_bias.restype = C.c_float
        d.orc_bias.argtypes = [C.c_float]
        d.orc_pack_signed.restype = C.c_uint32
        d.orc_pack_signed.argtypes = [C.c_int32]
        d.orc_hybridize.argtypes = [C.c_uint32, C.c_void_p]
        d.orc_normalize_frequencies.restype = C.c_int
        d.orc_normalize_frequencies.argtypes = [C.c_void_p, C.c_uint32]
        d.orc_alias_slot.restype = C.c_int
        d.orc_alias_slot.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_uint32, C.c_uint32]
        d.orc_hf_cluster_map.argtypes = [C.c_void_p, C.c_uint]
        d.orc_dct8x8.argtypes = [C.c_void_p, C.c_void_p]
        d.orc_encode_lf_group.restype = C.POINTER(_Result)
        d.orc_encode_lf_group.argtypes = [C.POINTER(C.c_void_p), C.c_ssize_t, C.c_ssize_t, C.c_int, C.c_int,
                                          C.c_size_t, C.c_size_t, C.c_uint, C.c_uint, C.POINTER(C.c_uint),
                                          C.POINTER(C.c_int)]
        d.orc_free_result.argtypes = [C.POINTER(_Result)]
        d.orc_hot_path_image.restype = C.c_int
        d.orc_hot_path_image.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_size_t, C.c_int,
                                         C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        _lib = d
    return _lib


def input_lut(size: int, need_linearize: bool) -> np.ndarray:
    out = np.zeros(size, np.uint16)
    lib().orc_build_input_lut(out.ctypes.data, size, int(need_linearize))
    return out


def bias_lut() -> np.ndarray:
    out = np.zeros(65536, np.float32)
    lib().orc_build_bias_lut(out.ctypes.data)
    return out


def hybridize(value: int):
    rec = np.zeros(1, SYMBOL_DTYPE)
    lib().orc_hybridize(value, rec.ctypes.data)
    return int(rec["token"][0]), int(rec["residue_bits"][0]), int(rec["residue"][0])


def normalize(freq) -> tuple:
    f = np.array(freq, np.uint32)
    uniq = lib().orc_normalize_frequencies(f.ctypes.data, len(f))
    return f, uniq


def alias_slot(freq, log_alphabet_size, unique, symbol, offset) -> int:
    f = np.ascontiguousarray(freq, np.uint32)
    return lib().orc_alias_slot(f.ctypes.data, len(f), log_alphabet_size, int(unique), symbol, offset)


def hf_cluster_map(num_presets: int) -> np.ndarray:
    out = np.zeros(1485 * num_presets, np.uint8)
    lib().orc_hf_cluster_map(out.ctypes.data, num_presets)
    return out


def dct8x8(block: np.ndarray) -> np.ndarray:
    src = np.ascontiguousarray(block, np.float32)
    out = np.zeros((8, 8), np.float32)
    lib().orc_dct8x8(src.ctypes.data, out.ctypes.data)
    return out


class LfResult:
    """numpy copies of every stage output of one LF group."""

    def __init__(self, r: _Result):
        self.width, self.height, self.vbw, self.vbh = r.width, r.height, r.vbw, r.vbh
        self.stride, self.gcols, self.grows, self.num_groups = r.stride, r.gcols, r.grows, r.num_groups
        shape = (3, self.vbh * 8, self.stride)
        n = 3 * self.vbh * 8 * self.stride
        self.xyb = np.ctypeslib.as_array(r.xyb, (n,)).reshape(shape).copy()
        self.dct = np.ctypeslib.as_array(r.dct, (n,)).reshape(shape).copy()
        self.quant = np.ctypeslib.as_array(r.quant, (n,)).reshape(shape).copy()
        self.dc = np.ctypeslib.as_array(r.dc, (3 * self.vbh * self.vbw,)).reshape(3, self.vbh, self.vbw).copy()
        self.nz = np.ctypeslib.as_array(r.nz, (self.num_groups * 1024 * 3,)).reshape(self.num_groups, 1024, 3).copy()
        self.num_symbols = r.num_symbols
        buf = (C.c_uint8 * (r.num_symbols * 8)).from_address(r.symbols) if r.num_symbols else b""
        self.symbols = np.frombuffer(bytes(buf), SYMBOL_DTYPE).copy()
        self.group_symbols = np.ctypeslib.as_array(r.group_symbols, (self.num_groups,)).astype(np.int64)
        self.cluster_from, self.cluster_to = r.cluster_from, r.cluster_to
        self.alphabet_size = np.array(r.alphabet_size[:], np.int64)
        self.freqs = np.array([list(row) for row in r.freqs], np.uint32)
        self.max_alphabet_size, self.log_alphabet_size = r.max_alphabet_size, r.log_alphabet_size
        self.stream = bytes(np.ctypeslib.as_array(r.stream, (r.stream_bytes,))) if r.stream_bytes else b""
        self.group_offset = np.ctypeslib.as_array(r.group_offset, (self.num_groups,)).astype(np.int64)
        self.group_bits = np.ctypeslib.as_array(r.group_bits, (self.num_groups,)).astype(np.int64)

    def group_stream(self, g: int) -> bytes:
        o = int(self.group_offset[g])
        return self.stream[o:o + (int(self.group_bits[g]) + 7) // 8]


def encode_lf_group_ptrs(ptrs, row_stride, pixel_stride, fmt, linear_light, width, height, preset, num_presets,
                         max_alphabet_size=0):
    arr = (C.c_void_p * 3)(*ptrs)
    mx = C.c_uint(max_alphabet_size)
    err = C.c_int(0)
    r = lib().orc_encode_lf_group(arr, row_stride, pixel_stride, fmt, linear_light, width, height, preset,
                                  num_presets, C.byref(mx), C.byref(err))
    if not r:
        raise RuntimeError(f"oracle failed with {err.value}")
    try:
        return LfResult(r.contents), mx.value
    finally:
        lib().orc_free_result(r)


def encode_lf_group(img: np.ndarray, tile_x: int = 0, tile_y: int = 0, *, linear_light: int = 0,
                    num_presets=None, preset=None, max_alphabet_size: int = 0):
    """Hot path on the 2048x2048 LF group (tile_x, tile_y) of an interleaved (H, W, 3) image."""
    h, w, _ = img.shape
    assert img.flags["C_CONTIGUOUS"]
    isz = img.dtype.itemsize
    x0, y0 = tile_x * 2048, tile_y * 2048
    lw, lh = min(2048, w - x0), min(2048, h - y0)
    lfx, lfy = -(-w // 2048), -(-h // 2048)
    if num_presets is None:
        num_presets = min(lfx * lfy, 256)
    if preset is None:
        preset = tile_y * lfx + tile_x
    base = img.ctypes.data + (y0 * w + x0) * 3 * isz
    return encode_lf_group_ptrs([base, base + isz, base + 2 * isz], 3 * w, 3, FMT[img.dtype], linear_light,
                                lw, lh, preset, num_presets, max_alphabet_size)


def hot_path_image(img: np.ndarray, linear_light: int = 0):
    """(total group-section bytes, FNV-1a checksum) of the hot path over a whole image."""
    h, w, _ = img.shape
    assert img.flags["C_CONTIGUOUS"]
    nbytes, cks = C.c_uint64(0), C.c_uint64(0)
    ret = lib().orc_hot_path_image(img.ctypes.data, FMT[img.dtype], w, h, linear_light, C.byref(nbytes), C.byref(cks))
    if ret:
        raise RuntimeError(f"oracle failed with {ret}")
    return nbytes.value, cks.value
