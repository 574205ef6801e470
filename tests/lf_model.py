"""A numpy model of what the GPU LF-group coder (hydrium_amd/csrc/hip/lf_coder.hip) must produce.

TEST INFRASTRUCTURE ONLY.  It restates the reference's LF-coefficient stream
(/root/reference/src/libhydrium/encoder.c:560-596 with entropy.c:427-524 for the symbol buffering
and entropy.c:664-707,1003-1021 for codes and write-out) as whole-array operations, taking the
code lengths from the product's host prefix coder (already pinned against the reference by
test_host_glue.py).  CPU tests check that the host splice around such a stream reproduces the
host-coded section; GPU tests check that the device produces exactly this model's output.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from hydrium_amd import build as hbuild

LF_CODES = 384
RUN_BASE = 16384

_d = None


def lib():
    global _d
    if _d is None:
        hbuild.build()
        d = C.CDLL(hbuild.HOSTTEST_PATH)
        d.hydt_code_lengths.restype = C.c_int
        d.hydt_code_lengths.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]
        d.hydt_lf_group.restype = C.c_int
        d.hydt_lf_group.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        d.hydt_lf_group_coded.restype = C.c_int
        d.hydt_lf_group_coded.argtypes = [C.c_size_t, C.c_size_t, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                          C.c_uint64, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        d.hydt_free.argtypes = [C.c_void_p]
        _d = d
    return _d


def _take(out, n):
    data = bytes((C.c_uint8 * n.value).from_address(out.value)) if n.value else b""
    lib().hydt_free(out)
    return data


def host_lf_group(dc: np.ndarray) -> bytes:
    """Byte-padded LFGroup section from LF ints dc[3][vbh][vbw] (X, Y, B) by the host coder."""
    dc = np.ascontiguousarray(dc, np.int32)
    out, n = C.c_void_p(0), C.c_size_t(0)
    ret = lib().hydt_lf_group(dc.ctypes.data, dc.shape[2], dc.shape[1], C.byref(out), C.byref(n))
    assert ret == 0, ret
    return _take(out, n)


def coded_lf_group(vbw, vbh, lengths, alphabet, run_pairs, bits: np.ndarray, bit_count) -> bytes:
    lengths = np.ascontiguousarray(lengths, np.uint8)
    bits = np.ascontiguousarray(bits, np.uint8)
    out, n = C.c_void_p(0), C.c_size_t(0)
    ret = lib().hydt_lf_group_coded(vbw, vbh, lengths.ctypes.data, alphabet, run_pairs, bits.ctypes.data, bit_count,
                                    C.byref(out), C.byref(n))
    assert ret == 0, ret
    return _take(out, n)


def host_code_lengths(freq: np.ndarray, max_depth: int = 15) -> np.ndarray:
    freq = np.ascontiguousarray(freq, np.uint32)
    lengths = np.zeros(len(freq), np.uint32)
    ret = lib().hydt_code_lengths(freq.ctypes.data, lengths.ctypes.data, len(freq), max_depth)
    assert ret == 0, ret
    return lengths


def compact_to_tokens(hist384: np.ndarray):
    """Histogram over the compact token space -> (full histogram, alphabet)."""
    nz = np.nonzero(hist384)[0]
    if len(nz) == 0:
        return np.zeros(1, np.uint32), 0
    top = int(nz[-1])
    alphabet = (top if top < 256 else RUN_BASE + top - 256) + 1
    full = np.zeros(alphabet, np.uint32)
    m = min(alphabet, 256)
    full[:m] = hist384[:m]
    if alphabet > RUN_BASE:
        k = alphabet - RUN_BASE
        full[RUN_BASE:] = hist384[256:256 + k]
    return full, alphabet


def lengths_for_hist(hist384: np.ndarray) -> np.ndarray:
    """Host code lengths, folded back into the compact token space."""
    full, alphabet = compact_to_tokens(hist384)
    out = np.zeros(LF_CODES, np.uint8)
    if alphabet <= 1:
        return out
    lens = host_code_lengths(full)
    m = min(alphabet, 256)
    out[:m] = lens[:m]
    if alphabet > RUN_BASE:
        out[256:256 + alphabet - RUN_BASE] = lens[RUN_BASE:]
    return out


def canonical_codes(lengths: np.ndarray) -> np.ndarray:
    """len << 16 | bit-reversed canonical code, shorter first, ties by token."""
    codes = np.zeros(len(lengths), np.uint32)
    nxt = 0
    for ln in range(1, 16):
        for i in np.nonzero(lengths == ln)[0]:
            code = nxt >> (32 - ln)
            rev = int(format(code, f"0{ln}b")[::-1], 2)
            codes[i] = (ln << 16) | rev
            nxt += 1 << (32 - ln)
    assert nxt in (0, 1 << 32), "incomplete code"
    return codes


def residuals(dc: np.ndarray) -> np.ndarray:
    """pack_signed(lf - clamped gradient), channels Y, X, B, raster (encoder.c:574-594)."""
    dc = np.asarray(dc, np.int64)
    out = []
    for c in (1, 0, 2):
        p = dc[c]
        h, w = p.shape
        W = np.zeros_like(p)
        W[:, 1:] = p[:, :-1]
        W[1:, 0] = p[:-1, 0]
        N = W.copy()
        N[1:, :] = p[:-1, :]
        NW = W.copy()
        NW[1:, 1:] = p[:-1, :-1]
        wrap = lambda a: ((a + (1 << 31)) % (1 << 32)) - (1 << 31)
        pred = np.clip(wrap(W + N - NW), np.minimum(W, N), np.maximum(W, N))
        d = wrap(p - pred)
        out.append(((d << 1) ^ (d >> 63)) & 0xFFFFFFFF)
    return np.concatenate([o.reshape(-1) for o in out]).astype(np.uint64)


def hybrid(v: np.ndarray):
    """hybrid-uint (7,1,1): token, residue bit count, residue."""
    v = v.astype(np.uint64)
    big = v >= 128
    L = np.zeros(len(v), np.int64)
    L[big] = np.floor(np.log2(v[big].astype(np.float64))).astype(np.int64)
    # guard against float rounding at powers of two
    L = np.where(big & ((np.uint64(1) << L.astype(np.uint64)) > v), L - 1, L)
    L = np.where(big & ((np.uint64(1) << (L + 1).astype(np.uint64)) <= v), L + 1, L)
    nb = np.where(big, L - 2, 0)
    res = np.where(big, (v >> np.uint64(1)) & ((np.uint64(1) << nb.astype(np.uint64)) - np.uint64(1)), 0)
    high = (v >> (nb + 1).astype(np.uint64)) & np.uint64(1)
    tok = np.where(big, 128 + (((nb - 5) << 2) | (high.astype(np.int64) << 1) | (v & np.uint64(1)).astype(np.int64)), v.astype(np.int64))
    return tok.astype(np.int64), nb.astype(np.int64), res.astype(np.uint64)


def emissions(v: np.ndarray):
    """Per position: literal flag and run length r (0 = no run pair), entropy.c:473-524."""
    n = len(v)
    head = np.ones(n, bool)
    head[1:] = v[1:] != v[:-1]
    idx = np.arange(n)
    start = np.maximum.accumulate(np.where(head, idx, 0))
    nxt = np.where(head, idx, n)
    end = np.minimum.accumulate(np.append(nxt[1:], n)[::-1])[::-1]  # first head after i
    off = idx - start
    c = off & 127
    chunk_start = idx - c
    r_chunk = np.minimum(127, end - chunk_start - 1)
    lit = (c == 0) | ((c <= 3) & (r_chunk <= 3))
    r = np.where((c == 0) & (r_chunk > 3), r_chunk, 0)
    return lit, r


def model(dc: np.ndarray):
    """(hist384, lengths384, alphabet, run_pairs, packed bits as uint8 array, bit_count)."""
    v = residuals(dc)
    lit, r = emissions(v)
    tok, nb, res = hybrid(v)
    hist = np.zeros(LF_CODES, np.uint32)
    np.add.at(hist, tok[lit], 1)
    np.add.at(hist, 256 + r[r > 0] - 3, 1)
    lengths = lengths_for_hist(hist)
    codes = canonical_codes(lengths)
    _, alphabet = compact_to_tokens(hist)
    # per-position bit strings
    e_lit = codes[np.where(lit, tok, 0)].astype(np.uint64)
    val = np.where(lit, (e_lit & np.uint64(0xFFFF)) | (res << (e_lit >> np.uint64(16))), np.uint64(0))
    ln = np.where(lit, (e_lit >> np.uint64(16)).astype(np.int64) + nb, 0)
    e_run = codes[np.where(r > 0, 256 + r - 3, 0)].astype(np.uint64)
    val = np.where(r > 0, val | ((e_run & np.uint64(0xFFFF)) << ln.astype(np.uint64)), val)
    ln = np.where(r > 0, ln + (e_run >> np.uint64(16)).astype(np.int64), ln)
    total = int(ln.sum())
    # expand to bits (LSB first); fine for test sizes
    keep = ln > 0
    val, ln = val[keep], ln[keep]
    offs = np.concatenate([[0], np.cumsum(ln)[:-1]]) if len(ln) else np.zeros(0, np.int64)
    bitarr = np.zeros(total + 7, np.uint8)
    maxlen = int(ln.max()) if len(ln) else 0
    for b in range(maxlen):
        m = ln > b
        bitarr[offs[m] + b] = ((val[m] >> np.uint64(b)) & np.uint64(1)).astype(np.uint8)
    packed = np.packbits(bitarr[: (total + 7) // 8 * 8], bitorder="little")
    return hist, lengths, alphabet, int((r > 0).sum()), packed, total
