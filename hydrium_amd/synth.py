"""Deterministic integer-only synthetic RGB images.

Three content classes drive tests and ``bench.py``:

* ``photo``  — multi-octave value noise (~1.5 bpp, ~0.5 symbols/px): the primary workload.
* ``smooth`` — ramp + mild hash noise (~0.2 bpp): best case for the entropy stage.
* ``noise``  — uniform hash noise (~14 bpp, ~2.9 symbols/px): worst case for the entropy stage.

Every sample is a pure function of (x, y, channel, seed) built from 32-bit hashing done in int64
with explicit masking, so the same code runs on numpy arrays (tests, CPU baseline) and on torch
tensors (``bench.py`` generates directly in HBM).  ``smooth``/``noise`` are defined with the
positional hash rather than the survey's serial xorshift stream so that they can be generated on
the device; only ``photo`` follows the survey's definition.
"""
from __future__ import annotations

import numpy as np

M32 = 0xFFFFFFFF


class _NP:
    int64 = np.int64

    @staticmethod
    def arange(n):
        return np.arange(n, dtype=np.int64)

    where = staticmethod(np.where)

    @staticmethod
    def clip(a, lo, hi):
        return np.clip(a, lo, hi)

    @staticmethod
    def stack_last(arrs):
        return np.stack(arrs, axis=-1)

    @staticmethod
    def cast(a, kind):
        return a.astype({"u8": np.uint8, "u16": np.uint16}[kind])


class _Torch:
    def __init__(self, device):
        import torch

        self.t = torch
        self.device = device

    def arange(self, n):
        return self.t.arange(n, dtype=self.t.int64, device=self.device)

    def where(self, c, a, b):
        t = self.t
        if not t.is_tensor(a):
            a = t.full_like(b if t.is_tensor(b) else c, a, dtype=t.int64)
        if not t.is_tensor(b):
            b = t.full_like(a, b)
        return t.where(c, a, b)

    def clip(self, a, lo, hi):
        return self.t.clamp(a, lo, hi)

    def stack_last(self, arrs):
        return self.t.stack(arrs, dim=-1)

    def cast(self, a, kind):
        t = self.t
        if kind == "u8":
            return a.to(t.uint8)
        # torch has no first-class uint16 arithmetic; int16 carries the same bit pattern
        return a.to(t.int32).to(t.int16).view(t.uint16) if hasattr(t, "uint16") else a.to(t.int16)


def _mix(h):
    h = h ^ (h >> 15)
    h = (h * 0x2C1B3C6D) & M32
    h = h ^ (h >> 12)
    h = (h * 0x297A2D39) & M32
    h = h ^ (h >> 15)
    return h


def _h32(x, y, c, s):
    h = ((x * 0x9E3779B1) & M32) ^ ((y * 0x85EBCA77) & M32) ^ ((c * 0xC2B2AE3D) & M32) ^ ((s * 0x27D4EB2F) & M32)
    return _mix(h)


def _tdiv(xp, a, b):
    """C-style (truncating) integer division by a positive constant."""
    neg = a < 0
    q = xp.where(neg, -a, a) // b
    return xp.where(neg, -q, q)


def _vnoise(xp, X, Y, c, octave, seed):
    if octave == 0:
        return _h32(X, Y, 16 * c, seed) & 0xFFFF
    cell = 1 << octave
    cx, cy = X >> octave, Y >> octave
    fx = (X & (cell - 1)) << (16 - octave)
    fy = (Y & (cell - 1)) << (16 - octave)
    ch = 16 * c + octave
    # hash the coarse lattice once, then gather the four corners per pixel
    cx0, cy0 = int(cx.min()), int(cy.min())
    nlx, nly = int(cx.max()) - cx0 + 2, int(cy.max()) - cy0 + 2
    lat = _h32((xp.arange(nlx) + cx0)[None, :], (xp.arange(nly) + cy0)[:, None], ch, seed) & 0xFFFF
    ix, iy = (cx - cx0)[0, :], (cy - cy0)[:, 0]
    v00 = lat[iy][:, ix]
    v10 = lat[iy][:, ix + 1]
    v01 = lat[iy + 1][:, ix]
    v11 = lat[iy + 1][:, ix + 1]
    top = v00 * (65536 - fx) + v10 * fx
    bot = v01 * (65536 - fx) + v11 * fx
    return (top * (65536 - fy) + bot * fy) >> 32


def _v16(xp, kind, X, Y, c, seed, cache=None):
    if kind == "photo":
        def acc(ch):
            a = None
            for octave in range(8, -1, -1):
                t = (octave + 1) * (_vnoise(xp, X, Y, ch, octave, seed) - 32768)
                a = t if a is None else a + t
            return a

        cache = {} if cache is None else cache
        if "lum" not in cache:
            cache["lum"] = _tdiv(xp, acc(7), 45)
        return xp.clip(32768 + _tdiv(xp, _tdiv(xp, acc(c), 45), 2) + cache["lum"], 0, 65535)
    r = _h32(X, Y, 64 + c, seed)
    if kind == "noise":
        return r & 0xFFFF
    if kind == "smooth":
        base = ((13 * X + 7 * Y * (c + 1)) >> 1) & 0xFFFF
        return xp.clip(base + (r % 2048) - 1024, 0, 65535)
    if kind == "black":
        return X * 0 + Y * 0
    if kind == "white":
        return X * 0 + Y * 0 + 65535
    if kind == "ramp":
        return ((X * 257 + Y * 131 * (c + 1)) & 0xFFFF)
    raise ValueError(f"unknown synthetic image kind {kind!r}")


KINDS = ("photo", "smooth", "noise", "black", "white", "ramp")


def make_image(kind: str, width: int, height: int, depth: int = 8, seed: int = 1234, *,
               x0: int = 0, y0: int = 0, device=None):
    """Return an interleaved (height, width, 3) RGB image, uint8 (``depth`` 8) or uint16 (16).

    ``x0``/``y0`` offset the sampling window so that a shard of a larger image can be generated
    on its own rank.  With ``device`` set the result is a torch tensor on that device.
    """
    xp = _NP if device is None else _Torch(device)
    X = (xp.arange(width) + x0)[None, :]
    Y = (xp.arange(height) + y0)[:, None]
    chans = []
    cache = {}
    for c in range(3):
        v = _v16(xp, kind, X, Y, c, seed, cache)
        if depth == 8:
            v = v >> 8
        chans.append(v + X * 0 + Y * 0)
    return xp.cast(xp.stack_last(chans), "u8" if depth == 8 else "u16")


def make_image_f32(kind: str, width: int, height: int, seed: int = 1234):
    """float32 variant in [0, 1] (numpy only): v16 / 65535 computed in float32."""
    img = make_image(kind, width, height, 16, seed).astype(np.float32)
    return (img * np.float32(1.0 / 65535.0)).astype(np.float32)
