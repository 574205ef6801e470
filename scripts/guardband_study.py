#!/usr/bin/env python3
"""Guard-band study (task 4): could K1 run a FAST 8x8 DCT (fused multiply-adds, butterflies)
and fall back to the exact ordered accumulation only for blocks where the faster arithmetic might change a
quantised integer?  CPU experiment on the oracle's stage planes; writes profiles/r03_guardband.txt.

For every block of an image this
  1. takes the oracle's exact XYB planes (the LUT front end is exact in either design: it is a pure function of
     16-bit indices, checked entry by entry on the device),
  2. recomputes the 2-D DCT two faster ways in binary32 —
       fma      the same 8-term sums with fused multiply-adds (8 instead of 15 operations per output)
       fly      even/odd butterfly: 36 operations per 8 outputs instead of 106, fused where possible
     (fused operations are emulated as round32(float64(a) * float64(b) + float64(c)): the product is exact in
     binary64, the double rounding of the sum is immaterial for error statistics),
  3. measures the error eps of the scaled value s = coefficient * weight * 5 against the exact path, and
  4. counts the blocks a guard band of width delta = 4 x (largest error seen on that content, per channel) would send
     to the exact path: blocks holding a coefficient with |s| >= 2 - delta whose distance to the next integer
     is below delta, or an LF int within delta of an integer — and the blocks where the fast arithmetic really
     changes an integer (every one of them must be inside the band).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hydrium_amd import synth  # noqa: E402
from oracle import binding as orc  # noqa: E402

F = np.float32
A, B, C, D, E, Fc, G = (F(0.17338), F(0.146984), F(0.0982119), F(0.0344874), F(0.16332), F(0.0676495), F(0.125))
DCT = np.array([[G] * 8,
                [A, B, C, D, -D, -C, -B, -A], [E, Fc, -Fc, -E, -E, -Fc, Fc, E], [B, -D, -A, -C, C, A, D, -B],
                [G, -G, -G, G, G, -G, -G, G], [C, -A, D, B, -B, -D, A, -C], [Fc, -E, E, -Fc, -Fc, E, -E, Fc],
                [D, -C, B, -A, A, -B, C, -D]], F)
ZZ = np.array([[0, 2, 3, 9, 10, 20, 21, 35], [1, 4, 8, 11, 19, 22, 34, 36], [5, 7, 12, 18, 23, 33, 37, 48],
               [6, 13, 17, 24, 32, 38, 47, 49], [14, 16, 25, 31, 39, 46, 50, 57], [15, 26, 30, 40, 45, 51, 56, 58],
               [27, 29, 41, 44, 52, 55, 59, 62], [28, 42, 43, 53, 54, 60, 61, 63]])  # [kv][kh]
QW = np.array([
    [1969, 1969, 1969, 1962, 1969, 1962, 1655, 1885, 1885, 1655, 1397, 1610, 1704, 1610, 1397, 1178, 1368, 1494, 1494, 1368, 1178,
     994, 1159, 1289, 1340, 1289, 1159, 994, 839, 980, 1104, 1178, 1178, 1104, 980, 839, 829, 941, 1023, 1054, 1023, 941, 829, 800,
     881, 928, 928, 881, 800, 755, 809, 829, 809, 755, 663, 731, 731, 663, 491, 524, 491, 349, 349, 239],
    [280, 280, 280, 279, 280, 279, 245, 271, 271, 245, 214, 239, 250, 239, 214, 188, 211, 226, 226, 211, 188, 164, 185, 201, 207, 201,
     185, 164, 144, 163, 178, 188, 188, 178, 163, 144, 143, 157, 168, 172, 168, 157, 143, 139, 150, 156, 156, 150, 139, 133, 140, 143,
     140, 133, 125, 129, 129, 125, 116, 118, 116, 107, 107, 98],
    [256, 147, 147, 85, 117, 85, 60, 78, 78, 60, 43, 56, 63, 56, 43, 43, 43, 48, 48, 43, 43, 42, 43, 43, 43, 43, 43, 42, 29, 41, 43, 43,
     43, 43, 41, 29, 29, 37, 43, 43, 43, 37, 29, 27, 33, 36, 36, 33, 27, 24, 27, 29, 27, 24, 20, 22, 22, 20, 15, 16, 15, 10, 10, 7]], F)
LF_SHIFT = np.array([8192, 1024, 512], F)


def fma(a, b, c):
    return (a.astype(np.float64) * np.float64(b) + c.astype(np.float64)).astype(F)


def dct8_exact(x):  # x[..., 8] -> [..., 8]; the reference's ordered accumulation, separate multiplies and additions
    out = []
    for k in range(8):
        acc = x[..., 0] * DCT[k, 0] if k else x[..., 0]
        for n in range(1, 8):
            acc = acc + (x[..., n] * DCT[k, n] if k else x[..., n])
        out.append(acc * G if k == 0 else acc)
    return np.stack(out, -1).astype(F)


def dct8_fma(x):
    out = []
    for k in range(8):
        if k == 0:
            acc = x[..., 0]
            for n in range(1, 8):
                acc = acc + x[..., n]
            out.append(acc * G)
            continue
        acc = x[..., 0] * DCT[k, 0]
        for n in range(1, 8):
            acc = fma(x[..., n], DCT[k, n], acc)
        out.append(acc)
    return np.stack(out, -1).astype(F)


def dct8_fly(x):
    x0, x1, x2, x3, x4, x5, x6, x7 = [x[..., i] for i in range(8)]
    s07, d07, s16, d16, s25, d25, s34, d34 = x0 + x7, x0 - x7, x1 + x6, x1 - x6, x2 + x5, x2 - x5, x3 + x4, x3 - x4
    e0, e1, e2, e3 = s07 + s34, s16 + s25, s07 - s34, s16 - s25
    X0, X4 = (e0 + e1) * G, (e0 - e1) * G
    X2 = fma(e3, Fc, e2 * E)
    X6 = fma(e3, -E, e2 * Fc)

    def odd(c0, c1, c2, c3):
        return fma(d34, c3, fma(d25, c2, fma(d16, c1, d07 * c0)))

    X1, X3, X5, X7 = odd(A, B, C, D), odd(B, -D, -A, -C), odd(C, -A, D, B), odd(D, -C, B, -A)
    return np.stack([X0, X1, X2, X3, X4, X5, X6, X7], -1).astype(F)


def blocks_of(plane):  # [H, W] -> [by, bx, row, col]
    h, w = plane.shape
    return plane.reshape(h // 8, 8, w // 8, 8).transpose(0, 2, 1, 3)


def dct2d(blk, f):
    rows = f(blk)                                # along columns of a row: [by, bx, row, kh]
    cols = f(rows.transpose(0, 1, 3, 2))         # along rows: [by, bx, kh, kv]
    return cols                                   # index [..., kh, kv]


def study(kind, depth, size, out):
    img = synth.make_image(kind, size, size, depth)
    res, _ = orc.encode_lf_group(np.ascontiguousarray(img))
    xyb = res.xyb[:, :size, :size]
    wq = np.zeros((3, 8, 8), F)
    for kv in range(8):
        for kh in range(8):
            wq[:, kh, kv] = QW[:, ZZ[kv, kh]]
    nblocks = (size // 8) ** 2
    exact = [dct2d(blocks_of(xyb[c]), dct8_exact) for c in range(3)]
    # sanity: our numpy restatement of the exact path gives the oracle's quantised integers
    q_exact = []
    for c in range(3):
        s = exact[c] * wq[c] * F(5)
        q = np.where(np.abs(s) >= 2, np.trunc(s), 0).astype(np.int32)
        q[..., 0, 0] = 0
        q_exact.append((s, q))
        oq = blocks_of(res.quant[c, :size, :size])  # oracle layout: block row kh, column kv
        assert (oq == q).all(), "numpy restatement of the exact path disagrees with the oracle"
    out.append(f"\n{kind} {size}x{size} RGB{depth}: {nblocks} blocks; non-zero quantised HF coefficients per block: "
               f"{sum(int((q != 0).sum()) for _, q in q_exact) / nblocks:.1f}")
    for name, f in (("fma", dct8_fma), ("fly", dct8_fly)):
        flagged = np.zeros(exact[0].shape[:2], bool)
        changed = np.zeros(exact[0].shape[:2], bool)
        rigorous = [None, None, None]
        escaped = 0
        line = []
        for c in range(3):
            fast = dct2d(blocks_of(xyb[c]), f)
            s_e, q_e = q_exact[c]
            s_f = fast * wq[c] * F(5)
            err = np.abs(s_f.astype(np.float64) - s_e.astype(np.float64))
            err[..., 0, 0] = 0
            eps = float(err.max())
            delta = 4 * eps
            q_f = np.where(np.abs(s_f) >= 2, np.trunc(s_f), 0).astype(np.int32)
            q_f[..., 0, 0] = 0
            frac = np.abs(s_f - np.round(s_f))
            near = (np.abs(s_f) >= 2 - delta) & (frac <= delta)
            near[..., 0, 0] = False
            lf_e = np.trunc(exact[c][..., 0, 0] * LF_SHIFT[c]).astype(np.int64)
            lf_s = fast[..., 0, 0] * LF_SHIFT[c]
            lf_err = float(np.abs(lf_s.astype(np.float64) - (exact[c][..., 0, 0] * LF_SHIFT[c]).astype(np.float64)).max())
            lf_near = np.abs(lf_s - np.round(lf_s)) <= 4 * lf_err
            lf_changed = np.trunc(lf_s).astype(np.int64) != lf_e
            # a band that needs no measurement: 5 roundings per 8-point pass (gamma_5), sum of |coefficients| of a row <= 1,
            # two passes -> |error of a coefficient| <= 10 u M with M the largest |sample| of the channel (SURVEY P8 ranges),
            # times weight x 5, plus the two roundings of the scaling itself
            u = 2.0 ** -24
            M = (0.028, 0.845, 0.39)[c]
            rig = 10 * u * M * wq[c] * 5 + 2 * u * np.abs(s_f)
            near_r = (np.abs(s_f) >= 2 - rig) & (frac <= rig)
            near_r[..., 0, 0] = False
            lf_near_r = np.abs(lf_s - np.round(lf_s)) <= 10 * u * M * LF_SHIFT[c] + u * np.abs(lf_s)
            rigorous[c] = near_r.any(axis=(2, 3)) | lf_near_r
            blk_flag = near.any(axis=(2, 3)) | lf_near
            blk_changed = (q_f != q_e).any(axis=(2, 3)) | lf_changed
            escaped += int((blk_changed & ~blk_flag).sum())
            flagged |= blk_flag
            changed |= blk_changed
            line.append(f"{'XYB'[c]}: eps {eps:.2e} (LF {lf_err:.2e})")
        out.append(f"  {name}: {'; '.join(line)}")
        out.append(f"       blocks whose integers change under the fast arithmetic: {changed.mean() * 100:.4f} %   "
                   f"blocks inside the guard band (4 x eps): {flagged.mean() * 100:.3f} %   changed but outside the band: {escaped}")
        rall = rigorous[0] | rigorous[1] | rigorous[2]
        out.append(f"       blocks inside the PROVABLE band (10 u M w 5 + 2 u |s|): {rall.mean() * 100:.2f} %  "
                   f"(X {rigorous[0].mean() * 100:.2f}, Y {rigorous[1].mean() * 100:.2f}, B {rigorous[2].mean() * 100:.2f}); "
                   f"strips of 32 blocks with at least one: {100 * rall.reshape(rall.shape[0], -1, 32).any(axis=2).mean():.1f} %")
    return out


def main():
    out = ["Guard-band study: exact ordered DCT vs fused / butterfly DCT in binary32 (scripts/guardband_study.py)",
           "eps = largest |s_fast - s_exact| seen, s = coefficient x weight x 5 (the value whose truncation is coded);",
           "a block is 'inside the guard band' if some |s| >= 2 - 4 eps lies within 4 eps of an integer (or an LF int does)."]
    size = int(os.environ.get("GB_SIZE", "1024"))
    for kind, depth in (("photo", 16), ("photo", 8), ("smooth", 8), ("noise", 8)):
        study(kind, depth, size, out)
    text = "\n".join(out) + "\n"
    print(text)
    with open(os.path.join(ROOT, "profiles", "r03_guardband.txt"), "w") as f:
        f.write(text)


if __name__ == "__main__":
    main()
