"""CPU: the field writers the device-side frame assembler runs (csrc/hip/hydk_sections.h) against the
host functions they restate (csrc/host/prefix.c, frame.c — themselves held to the reference by
test_host_glue.py).  The header compiles for both sides; here it is the host build of it
(libhydrium_hosttest.so).  The GPU tests hold the kernels that call it to whole reference files."""
import ctypes as C

import numpy as np
import pytest

from hydrium_amd import build as hbuild

LF_CODES, RUN_BASE = 384, 16384


@pytest.fixture(scope="module")
def lib():
    hbuild.build()
    d = C.CDLL(hbuild.HOSTTEST_PATH)
    for name in ("hydt_lf_head_host", "hydt_lf_head_sections", "hydt_lf_head_wave"):
        getattr(d, name).restype = C.c_int
        getattr(d, name).argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]
    for name in ("hydt_ans_distribution_host", "hydt_ans_distribution_sections"):
        getattr(d, name).restype = C.c_int
        getattr(d, name).argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]
    d.hydt_toc_entry_check.restype = C.c_int
    d.hydt_toc_entry_check.argtypes = [C.c_uint64]
    d.hydt_small_code_lengths.restype = C.c_int
    d.hydt_small_code_lengths.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]
    d.hydt_code_lengths.restype = C.c_int
    d.hydt_code_lengths.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]
    return d


def _bits(fn, *args):
    out = np.zeros(8192, np.uint8)
    n = C.c_uint64(0)
    ret = fn(*args, out.ctypes.data, out.size, C.byref(n))
    return ret, int(n.value), out[: (int(n.value) + 7) // 8].tobytes()


def _lengths_for(lib, hist):
    """compact histogram [384] -> (lengths [384] uint8, alphabet0, run_pairs) as the LF coder's code construction gives them"""
    hist = np.asarray(hist, np.uint32)
    used = np.nonzero(hist)[0]
    top = int(used.max())
    n = (top if top < 256 else RUN_BASE + top - 256) + 1
    full = np.zeros(n, np.uint32)
    full[: min(n, 256)] = hist[: min(n, 256)]
    if n > RUN_BASE:
        full[RUN_BASE:n] = hist[256 : 256 + n - RUN_BASE]
    lens = np.zeros(n, np.uint32)
    assert lib.hydt_code_lengths(full.ctypes.data, lens.ctypes.data, n, 15) == 0
    compact = np.zeros(LF_CODES, np.uint8)
    compact[: min(n, 256)] = lens[: min(n, 256)]
    if n > RUN_BASE:
        compact[256 : 256 + n - RUN_BASE] = lens[RUN_BASE:n]
    return compact, n, int(hist[256:].sum())


def _hist_cases():
    rng = np.random.default_rng(7)
    cases = []
    for k in (1, 2, 3, 4, 5, 8, 17, 60, 200):  # used literal tokens: the simple forms (<= 4) and the complex one
        h = np.zeros(LF_CODES, np.uint32)
        idx = rng.choice(228, size=k, replace=False)
        h[idx] = rng.integers(1, 5000, size=k)
        cases.append(h)
    for k in (1, 2, 3, 4):  # skewed simple forms: lengths (1,2,2), (1,2,3,3), (2,2,2,2)
        h = np.zeros(LF_CODES, np.uint32)
        h[rng.choice(100, size=k, replace=False)] = [1000, 10, 3, 1][:k]
        cases.append(h)
    h = np.zeros(LF_CODES, np.uint32)
    h[:8] = 5  # eight equally likely tokens: a single code length, the 18-symbol meta code degenerates
    cases.append(h)
    for runs in (1, 3, 40, 124):  # run tokens: the alphabet jumps to 16384 + r, a zero run of ~16000 lengths in between
        h = np.zeros(LF_CODES, np.uint32)
        h[rng.choice(228, size=30, replace=False)] = rng.integers(1, 3000, size=30)
        h[256 + rng.choice(124, size=runs, replace=False)] = rng.integers(1, 200, size=runs)
        cases.append(h)
    h = np.zeros(LF_CODES, np.uint32)
    h[0] = 7
    h[256 + 5] = 2  # two tokens, one of them a run token
    cases.append(h)
    h = np.ones(LF_CODES, np.uint32)  # everything in use (352 real tokens: 228 literals + 124 run tokens)
    h[228:256] = 0
    h[256 + 124 :] = 0
    cases.append(h * rng.integers(1, 1000, size=LF_CODES).astype(np.uint32))
    return cases


@pytest.mark.parametrize("case", range(len(_hist_cases())))
def test_lf_stream_header(lib, case):
    hist = _hist_cases()[case]
    lens, n, pairs = _lengths_for(lib, hist)
    for run_pairs in {pairs, 0 if n <= RUN_BASE else pairs}:
        a = _bits(lib.hydt_lf_head_host, lens.ctypes.data, n, run_pairs)
        b = _bits(lib.hydt_lf_head_sections, lens.ctypes.data, n, run_pairs)
        w = _bits(lib.hydt_lf_head_wave, lens.ctypes.data, n, run_pairs)
        assert a[0] == 0 and b[0] == 0 and w[0] == 0
        assert a[1] == b[1], f"bit counts differ: host {a[1]} sections {b[1]}"
        assert a[2] == b[2]
        assert w == a, "the wavefront form of the header writer differs from the host's"
        assert b[1] <= 640 * 32  # the assembler's per-slot scratch


def test_lf_stream_header_random(lib):
    rng = np.random.default_rng(11)
    for _ in range(150):
        h = np.zeros(LF_CODES, np.uint32)
        k = int(rng.integers(1, 228))
        h[rng.choice(228, size=k, replace=False)] = (rng.pareto(1.2, size=k) * 20 + 1).astype(np.uint32)
        if rng.random() < 0.6:
            r = int(rng.integers(1, 124))
            h[256 + rng.choice(124, size=r, replace=False)] = (rng.pareto(1.5, size=r) * 5 + 1).astype(np.uint32)
        lens, n, pairs = _lengths_for(lib, h)
        a = _bits(lib.hydt_lf_head_host, lens.ctypes.data, n, pairs)
        b = _bits(lib.hydt_lf_head_sections, lens.ctypes.data, n, pairs)
        w = _bits(lib.hydt_lf_head_wave, lens.ctypes.data, n, pairs)
        assert a == b == w and a[0] == 0


def test_small_code_lengths(lib):
    rng = np.random.default_rng(3)
    for trial in range(3000):
        f = np.zeros(18, np.uint32)
        k = int(rng.integers(1, 19))
        kind = trial % 5
        if kind == 0:
            vals = rng.integers(1, 6, size=k)           # many ties
        elif kind == 1:
            vals = rng.integers(1, 400, size=k)
        elif kind == 2:
            vals = 1 << rng.permutation(18)[:k]         # geometric: the depth limit (5) has to flatten the tree
        elif kind == 3:
            vals = np.full(k, int(rng.integers(1, 50)))  # all equal: merged nodes tie with one another
        else:
            vals = (rng.pareto(0.7, size=k) * 3 + 1).astype(np.int64)
        f[rng.choice(18, size=k, replace=False)] = np.minimum(vals, 1 << 20)
        a, b = np.zeros(18, np.uint32), np.zeros(18, np.uint32)
        ra = lib.hydt_code_lengths(f.ctypes.data, a.ctypes.data, 18, 5)
        rb = lib.hydt_small_code_lengths(f.ctypes.data, b.ctypes.data, 18, 5)
        assert (ra == 0) == (rb == 0), (f, ra, rb)
        if ra == 0:
            assert (a == b).all(), (f, a, b)


def _normalised(rng, alphabet, used):
    f = np.zeros(128, np.uint32)
    idx = rng.choice(alphabet, size=used, replace=False)
    idx[0] = alphabet - 1  # the alphabet ends at its last used token
    w = rng.pareto(1.0, size=used) + 0.01
    v = np.maximum(1, np.floor(w / w.sum() * 4096)).astype(np.int64)
    v[np.argmax(v)] += 4096 - v.sum()
    if v.min() < 1:
        return None
    f[idx] = v.astype(np.uint32)
    return f


def test_ans_distribution(lib):
    rng = np.random.default_rng(5)
    cases = [(np.zeros(128, np.uint32), 0)]
    for a in (1, 2, 3, 5, 36, 72):
        f = np.zeros(128, np.uint32)
        f[a - 1] = 4096  # one symbol
        cases.append((f, a))
    for a, (i, j, p) in ((2, (0, 1, 100)), (9, (2, 8, 4000)), (36, (0, 35, 1))):
        f = np.zeros(128, np.uint32)
        f[i], f[j] = p, 4096 - p  # two symbols
        cases.append((f, a))
    f = np.zeros(128, np.uint32)
    f[0], f[3], f[4] = 2048, 2047, 1  # first two do not add up: the general form
    cases.append((f, 5))
    for _ in range(300):
        a = int(rng.integers(3, 73))
        f = _normalised(rng, a, int(rng.integers(3, a + 1)))
        if f is not None:
            cases.append((f, a))
    for f, a in cases:
        x = _bits(lib.hydt_ans_distribution_host, f.ctypes.data, a)
        y = _bits(lib.hydt_ans_distribution_sections, f.ctypes.data, a)
        assert x == y and x[0] == 0, (a, f[:a])


def test_toc_entries(lib):
    edges = [0, 1, 1023, 1024, 1025, 17407, 17408, 17409, 4211711, 4211712, 4211713, (1 << 30) + 4211711, (1 << 30) + 4211712,
             (1 << 32) - 1, 1 << 32, 1 << 40]
    rng = np.random.default_rng(1)
    for v in edges + [int(x) for x in rng.integers(0, 1 << 31, size=2000)]:
        assert lib.hydt_toc_entry_check(v) == 0, v
