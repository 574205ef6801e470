"""Drive the product's host-side frame glue (libhydrium_hosttest.so) with stage results computed by
the oracle.  CPU only: this checks headers, TOC, LF groups and HFGlobal against the reference
without a GPU; the GPU tests check the same glue fed by the HIP kernels."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from hydrium_amd import api, build as hbuild
from oracle import binding as orc

MAXC, ALPHA, GPL = 9, 128, 64


def _lib():
    hbuild.build()
    d = C.CDLL(hbuild.HOSTTEST_PATH)
    d.hydt_frame_from_stages.restype = C.c_int
    d.hydt_frame_from_stages.argtypes = [
        C.POINTER(api.HYDImageMetadata), C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.POINTER(C.c_void_p), C.c_void_p,
        C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p),
        C.POINTER(C.c_size_t), C.POINTER(C.c_char_p)]
    d.hydt_free.argtypes = [C.c_void_p]
    d.hydamd_frame_from_streams.restype = C.c_int
    d.hydamd_frame_from_streams.argtypes = [
        C.POINTER(api.HYDImageMetadata), C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
        C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_void_p),
        C.POINTER(C.c_size_t), C.POINTER(C.c_char_p)]
    d.hydamd_free.argtypes = [C.c_void_p]
    return d


class LfStream(C.Structure):  # include/hydrium_amd.h HydAmdLfStream
    _fields_ = [("lengths", C.c_void_p), ("alphabet", C.c_uint32), ("run_pairs", C.c_uint32), ("bits", C.c_void_p),
                ("bit_count", C.c_uint64)]


_d = None


def frame_from_stages(md, write_header, is_last, tiles, results, max_alphabet, icc=None, coded_lf=False) -> bytes:
    """coded_lf: hand the glue LF coefficient STREAMS (as the GPU LF coder produces them; here from the
    numpy model of tests/lf_model.py) through hydamd_frame_from_streams instead of LF ints."""
    global _d
    if _d is None:
        _d = _lib()
    n = len(results)
    tile_xy = np.array(tiles, np.uint32).reshape(-1)
    dcs = [np.ascontiguousarray(r.dc, np.int32) for r in results]
    dcp = (C.c_void_p * n)(*[a.ctypes.data for a in dcs])
    freq = np.zeros((n, MAXC, ALPHA), np.uint32)
    alpha = np.zeros((n, MAXC), np.uint32)
    bits = np.zeros((n, GPL), np.uint32)
    for s, r in enumerate(results):
        ncl = r.cluster_to - r.cluster_from
        freq[s, :ncl] = r.freqs[r.cluster_from:r.cluster_to]
        alpha[s, :ncl] = r.alphabet_size[r.cluster_from:r.cluster_to]
        bits[s, :r.num_groups] = r.group_bits
    payload = b"".join(r.stream for r in results)
    out, out_len, err = C.c_void_p(0), C.c_size_t(0), C.c_char_p(None)
    if coded_lf:
        import lf_model

        keep, arr = [], (LfStream * n)()
        for i, a in enumerate(dcs):
            _, lengths, alphabet, pairs, packed, nbits = lf_model.model(a)
            lengths = np.ascontiguousarray(lengths, np.uint8)
            packed = np.ascontiguousarray(packed, np.uint8)
            keep.append((lengths, packed))
            arr[i] = LfStream(lengths.ctypes.data, alphabet, pairs, packed.ctypes.data if nbits else None, nbits)
        ret = _d.hydamd_frame_from_streams(C.byref(md), int(write_header), int(is_last), n, tile_xy.ctypes.data, arr,
                                           freq.ctypes.data, alpha.ctypes.data, bits.ctypes.data, max_alphabet, payload,
                                           len(payload), icc, len(icc) if icc else 0, C.byref(out), C.byref(out_len),
                                           C.byref(err))
        if ret:
            raise RuntimeError(f"glue failed {ret}: {err.value}")
        data = bytes((C.c_uint8 * out_len.value).from_address(out.value))
        _d.hydamd_free(out)
        return data
    ret = _d.hydt_frame_from_stages(C.byref(md), int(write_header), int(is_last), n, tile_xy.ctypes.data, dcp,
                                    freq.ctypes.data, alpha.ctypes.data, bits.ctypes.data, max_alphabet, payload,
                                    len(payload), icc, len(icc) if icc else 0, C.byref(out), C.byref(out_len),
                                    C.byref(err))
    if ret:
        raise RuntimeError(f"glue failed {ret}: {err.value}")
    data = bytes((C.c_uint8 * out_len.value).from_address(out.value))
    _d.hydt_free(out)
    return data


def encode_with_oracle_stages(img: np.ndarray, shift_x=-1, shift_y=-1, order=None, icc=None, linear_light=0,
                              coded_lf=False) -> bytes:
    """Whole codestream: hot-path stages from the oracle, everything else from the product's host glue."""
    h, w, _ = img.shape
    md = api.HYDImageMetadata(w, h, linear_light, shift_x, shift_y)
    one_frame = shift_x < 0 or shift_y < 0
    tw, th = api.tile_dims(w, h, shift_x, shift_y)
    ntx, nty = -(-w // tw), -(-h // th)
    tiles = [(tx, ty) for ty in range(nty) for tx in range(ntx)] if order is None else list(order)
    isz = img.dtype.itemsize
    fmt = orc.FMT[img.dtype]

    def stage(tx, ty, preset, num_presets, mx):
        x0, y0 = tx * tw, ty * th
        p = img.ctypes.data + (y0 * w + x0) * 3 * isz
        return orc.encode_lf_group_ptrs([p, p + isz, p + 2 * isz], 3 * w, 3, fmt, linear_light,
                                        min(tw, w - x0), min(th, h - y0), preset, num_presets, mx)

    if one_frame:
        results, mx = [], 0
        for tx, ty in tiles:
            r, mx = stage(tx, ty, ty * ntx + tx, ntx * nty, mx)
            results.append(r)
        return frame_from_stages(md, True, True, tiles, results, mx, icc, coded_lf)
    out = b""
    for i, (tx, ty) in enumerate(tiles):
        last = (tx == ntx - 1 and ty == nty - 1) if order is None else i == len(tiles) - 1
        r, mx = stage(tx, ty, 0, 1, 0)
        out += frame_from_stages(md, i == 0, last, [(tx, ty)], [r], mx, None, coded_lf)
    return out
