/*
 * lf_huffman.h — the LF coder's prefix-code construction (one wavefront per LF group) and the scratch
 * record its kernels share.  Included by lf_coder.hip (the LF coder proper) and by kernels.hip, whose
 * entropy-stage launch carries the code construction of the same LF groups in extra workgroups: both
 * are one wavefront per LF group, both take hundreds of microseconds of serial work, neither needs the
 * other.  Everything here is internal linkage (include it inside an anonymous namespace).
 */
#ifndef HYDK_LF_HUFFMAN_H_
#define HYDK_LF_HUFFMAN_H_

constexpr int kLfThreads = 256;
constexpr int kLfWaves = kLfThreads / 64;
constexpr int kScanSpan = kLfThreads * 4;       /* values a token workgroup looks at */
constexpr int kAhead = 128;                     /* a run chunk is at most 128 values */
constexpr int kEmitSpan = kScanSpan - kAhead;   /* values it decides */
constexpr int kMaxWindows = (HYDK_LF_SYMBOLS + kEmitSpan - 1) / kEmitSpan; /* 220 windows of 896 values */

/* per-LF-group scratch between the kernels */
struct LfWork {
    uint32_t codes[HYDK_LF_CODES];               /* length << 16 | bit-reversed code per compact token */
    uint32_t win_residue_bits[kMaxWindows];      /* residue bits the window's literals carry */
    uint32_t win_off[kMaxWindows];               /* bits in front of each window */
    uint16_t win_hist[kMaxWindows][HYDK_LF_CODES]; /* tokens of each window: with the code lengths, its size in bits */
};
/* wave-level ordering of LDS traffic inside one wavefront (lock-step execution + in-order LDS make
 * the data visible; this only pins the compiler) */
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* ==========================================================================================
 * k_lf_huffman: grid = LF groups, block = 64 (one wavefront)
 *
 * Slot space of the reference's node array for an alphabet of n tokens: leaves 0..n-1, merged node
 * k at n + k; round k settles slots 2k and 2k+1.  With at most 352 weighted tokens there are at
 * most 351 rounds, so only these slots ever hold or receive a weighted node:
 *   [0, 768)              low leaves and the settle targets 2k, 2k+1 <= 703
 *   [16384, 16512)        run tokens                                  -> compact 768 + (s - 16384)
 *   [n, n + 384)          merged nodes                                -> compact 896 + (s - n)
 * The map is monotonic, so "ascending slot order" (the reference's visiting order, which decides
 * ties between merged nodes) is ascending compact order.
 * ======================================================================================== */
constexpr int kRunLo = 768, kMergedLo = 896; /* compact slots run up to 896 + 384 = 1280 < 2048 (11 bits of the entry meta) */
constexpr int kMaxDepth = 15;

__device__ __forceinline__ int lf_compact(int slot, int n) {
    return slot >= n ? kMergedLo + (slot - n) : slot >= HYDK_LF_RUN_BASE ? kRunLo + (slot - HYDK_LF_RUN_BASE) : slot;
}

/* A candidate ("entry") is a tree root still waiting to be merged.  Entries live in registers, 6 per
 * lane (384 = the number of leaves; every merge retires two candidates and creates one, which
 * takes over a retired entry).
 *   meta: bits 0-10 slot (compact), 11-20 "who" (the node's identity: compact token for a leaf,
 *         384 + k for the node merged in round k), 21-25 subtree height
 *   key:  weight << 12 | order, where order = token for a leaf and 0x800 | (2047 - slot) for a merged
 *         node: ascending key is exactly the reference's selection order (entropy.c:577-581: weight,
 *         then leaves before merged nodes, leaves by token, merged nodes by descending slot).
 *         0xFFFFFFFF marks a retired entry.  Weights stay below 2^20 (<= 2 symbols per LF value). */
constexpr int kEntries = HYDK_LF_CODES / 64;
constexpr uint32_t kDead = 0xFFFFFFFFu;
#define M_SLOT(m) ((m) & 2047u)
#define M_WHO(m) (((m) >> 11) & 1023u)
#define M_DEEP(m) (((m) >> 21) & 31u)
#define M_MAKE(slot, who, deep) ((uint32_t)(slot) | ((uint32_t)(who) << 11) | ((uint32_t)(deep) << 21))

__device__ __forceinline__ uint32_t lf_key(uint32_t weight, uint32_t who, uint32_t slot) {
    return (weight << 12) | (who >= (uint32_t)HYDK_LF_CODES ? 0x800u | (2047u - slot) : who);
}

#define LF_DPP(v, ctrl, rmask) ((uint32_t)__builtin_amdgcn_update_dpp((int)(v), (int)(v), ctrl, rmask, 0xF, false))

/* minimum over the wavefront, returned to every lane: row_shr 1/2/4/8, row_bcast 15/31, read lane 63 */
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    uint32_t t;
    t = LF_DPP(v, 0x111, 0xF);
    v = t < v ? t : v;
    t = LF_DPP(v, 0x112, 0xF);
    v = t < v ? t : v;
    t = LF_DPP(v, 0x114, 0xF);
    v = t < v ? t : v;
    t = LF_DPP(v, 0x118, 0xF);
    v = t < v ? t : v;
    t = LF_DPP(v, 0x142, 0xA);
    v = t < v ? t : v;
    t = LF_DPP(v, 0x143, 0xC);
    v = t < v ? t : v;
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

/* the value `v` of the one lane where `has` is set (0 when there is none) */
__device__ __forceinline__ uint32_t pick_lane(bool has, uint32_t v, bool &any) {
    const unsigned long long m = __ballot(has);
    any = m != 0;
    return any ? (uint32_t)__builtin_amdgcn_readlane((int)v, __builtin_ctzll(m)) : 0u;
}

__device__ __forceinline__ int clog2_i(int v) { return v <= 1 ? 0 : 32 - __clz(v - 1); }

struct LfHuffScratch {
    uint16_t par[2 * HYDK_LF_CODES];   /* round in which the node (leaf, then merged) was settled as a child */
    uint16_t depth[2 * HYDK_LF_CODES]; /* merged nodes above it */
    uint32_t cnt[16], first[16];
    uint32_t err;
};

/* run by ONE wavefront (lane = 0..63); hist / codes may live in LDS or in global memory */
__device__ __forceinline__ void lf_huffman_wave(const uint32_t *hist, uint32_t *codes, HydkLfStream *st, LfHuffScratch &S,
                                                int lane) {
    const unsigned long long ltmask = (1ull << lane) - 1ull;
    uint16_t *s_par = S.par, *s_depth = S.depth;
    uint32_t *s_cnt = S.cnt, *s_first = S.first;
    uint32_t &s_err = S.err;

    if (lane < 16)
        s_cnt[lane] = 0;
    if (lane == 0)
        s_err = 0;

    uint32_t ek[kEntries], em[kEntries], f6[kEntries];
    int maxidx = -1, live0 = 0;
    uint32_t pairs = 0, total = 0;
#pragma unroll
    for (int t = 0; t < kEntries; t++) {
        const int ci = t * 64 + lane;
        const uint32_t f = hist[ci];
        const uint32_t sl = ci < 256 ? ci : kRunLo + (ci - 256);
        f6[t] = f;
        em[t] = M_MAKE(sl, ci, 0);
        ek[t] = f ? lf_key(f, ci, sl) : kDead;
        if (f)
            maxidx = ci;
        live0 += (int)__popcll(__ballot(f != 0));
        total += f;
        if (ci >= 256)
            pairs += f;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const int t = __shfl_xor(maxidx, d);
        maxidx = t > maxidx ? t : maxidx;
        pairs += __shfl_xor(pairs, d);
        const uint32_t hi = __shfl_xor(total, d);
        total = total + hi < total ? 0xFFFFFFFFu : total + hi; /* saturating */
    }
    const int n = maxidx < 0 ? 0 : (maxidx < 256 ? maxidx : HYDK_LF_RUN_BASE + maxidx - 256) + 1;

    uint32_t err = live0 == 0 ? 1u : total >= (1u << 20) ? 6u : 0u;
    int merges = 0;
    for (int k = 0; k + 1 < n && !err; k++) {
        const uint32_t limit = (uint32_t)(kMaxDepth - clog2_i(live0 - k) + 1);
        const uint32_t c2k = (uint32_t)lf_compact(2 * k, n), c2k1 = (uint32_t)lf_compact(2 * k + 1, n);
        /* the two smallest candidates; those whose subtree is too tall for the depth limit sit the round out */
        uint32_t b1 = kDead, b2 = kDead, m1 = 0, m2 = 0;
#pragma unroll
        for (int t = 0; t < kEntries; t++) {
            const uint32_t key = M_DEEP(em[t]) < limit ? ek[t] : kDead;
            const bool lt1 = key < b1, lt2 = key < b2;
            b2 = lt1 ? b1 : lt2 ? key : b2;
            m2 = lt1 ? m1 : lt2 ? em[t] : m2;
            b1 = lt1 ? key : b1;
            m1 = lt1 ? em[t] : m1;
        }
        const uint32_t first = wave_min_u32(b1);
        if (first == kDead) {
            err = 2;
            break;
        }
        bool any;
        const uint32_t fm = pick_lane(b1 == first, m1, any);
        const uint32_t c = b1 == first ? b2 : b1, cm = b1 == first ? m2 : m1;
        const uint32_t second = wave_min_u32(c);
        const uint32_t sm = pick_lane(second != kDead && c == second, cm, any);
        const uint32_t f1 = M_SLOT(fm);

        /* "swap the pick with slot 2k": the pick settles there; whatever weighted node sat in slot 2k
         * moves to the pick's old slot */
        uint32_t ak = kDead, am = 0;
#pragma unroll
        for (int t = 0; t < kEntries; t++) {
            const bool hit = ek[t] != kDead && M_SLOT(em[t]) == c2k;
            ak = hit ? ek[t] : ak;
            am = hit ? em[t] : am;
        }
        bool has_a;
        const uint32_t a_k = pick_lane(ak != kDead, ak, has_a), a_m = pick_lane(ak != kDead, am, any);
#pragma unroll
        for (int t = 0; t < kEntries; t++) {
            const uint32_t sl = M_SLOT(em[t]);
            const bool alive = ek[t] != kDead;
            const bool at2k = alive && sl == c2k, atf1 = alive && sl == f1 && sl != c2k;
            const uint32_t moved_m = sl | (a_m & ~2047u);
            ek[t] = at2k ? kDead : atf1 ? (has_a ? lf_key(a_k >> 12, M_WHO(a_m), sl) : kDead) : ek[t];
            em[t] = atf1 && has_a ? moved_m : em[t];
        }
        if (second == kDead)
            break; /* a single tree is left */
        uint32_t f2 = M_SLOT(sm);
        if (f2 == c2k)
            f2 = f1; /* it was just moved out of slot 2k */
        uint32_t bk = kDead, bm = 0;
#pragma unroll
        for (int t = 0; t < kEntries; t++) {
            const bool hit = ek[t] != kDead && M_SLOT(em[t]) == c2k1;
            bk = hit ? ek[t] : bk;
            bm = hit ? em[t] : bm;
        }
        bool has_b;
        const uint32_t b_k = pick_lane(bk != kDead, bk, has_b), b_m = pick_lane(bk != kDead, bm, any);
        /* same for slot 2k+1; the entry this retires is reused for the merged node, which enters as
         * the highest slot so far */
        const uint32_t hf = M_DEEP(fm), hs = M_DEEP(sm);
        const uint32_t pslot = (uint32_t)(kMergedLo + k), pwho = (uint32_t)(HYDK_LF_CODES + k);
        const uint32_t parent_m = M_MAKE(pslot, pwho, 1u + (hf > hs ? hf : hs));
        const uint32_t parent_k = lf_key((first >> 12) + (second >> 12), pwho, pslot);
#pragma unroll
        for (int t = 0; t < kEntries; t++) {
            const uint32_t sl = M_SLOT(em[t]);
            const bool alive = ek[t] != kDead;
            const bool at2k1 = alive && sl == c2k1, atf2 = alive && sl == f2 && sl != c2k1;
            const bool takes_b = atf2 && has_b, becomes_parent = at2k1 || (atf2 && !has_b);
            ek[t] = becomes_parent ? parent_k : takes_b ? lf_key(b_k >> 12, M_WHO(b_m), sl) : ek[t];
            em[t] = becomes_parent ? parent_m : takes_b ? (sl | (b_m & ~2047u)) : em[t];
        }
        if (lane == 0) {
            s_par[M_WHO(fm)] = (uint16_t)k;
            s_par[M_WHO(sm)] = (uint16_t)k;
        }
        merges = k + 1;
    }
    if (!err && live0 - merges != 1)
        err = 3; /* the depth limit left more than one tree */
    wave_sync();

    /* code length of a leaf = merged nodes above it; the node merged last is the root */
    if (lane == 0 && merges > 0) {
        s_depth[HYDK_LF_CODES + merges - 1] = 0;
        for (int m = merges - 2; m >= 0; m--)
            s_depth[HYDK_LF_CODES + m] = (uint16_t)(s_depth[HYDK_LF_CODES + s_par[HYDK_LF_CODES + m]] + 1);
    }
    wave_sync();
    uint32_t len6[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        const int ci = j * 64 + lane;
        uint32_t len = 0;
        if (f6[j] && merges > 0 && !err)
            len = (uint32_t)s_depth[HYDK_LF_CODES + s_par[ci]] + 1u;
        if (len > (uint32_t)kMaxDepth) {
            err = 4;
            len = 0;
        }
        len6[j] = len;
        if (len)
            atomicAdd(&s_cnt[len], 1u);
    }
    wave_sync();
    /* canonical codes: shorter first, ties by token (entropy.c:664-707) */
    if (lane == 0) {
        unsigned long long next = 0;
        for (int L = 1; L <= kMaxDepth; L++) {
            s_first[L] = (uint32_t)(next >> (32 - L));
            next += (unsigned long long)s_cnt[L] << (32 - L);
        }
        if (next && next != (1ull << 32))
            atomicOr(&s_err, 5u);
    }
    if (err)
        atomicOr(&s_err, err);
    wave_sync();
    uint32_t run[kMaxDepth + 1];
#pragma unroll
    for (int L = 1; L <= kMaxDepth; L++)
        run[L] = s_first[L];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        const int ci = j * 64 + lane;
        uint32_t code = 0;
#pragma unroll
        for (int L = 1; L <= kMaxDepth; L++) {
            const unsigned long long m = __ballot(len6[j] == (uint32_t)L);
            if (len6[j] == (uint32_t)L)
                code = run[L] + (uint32_t)__popcll(m & ltmask);
            run[L] += (uint32_t)__popcll(m);
        }
        const uint32_t len = len6[j];
        codes[ci] = len ? (len << 16) | (__brev(code) >> (32u - len)) : 0u;
        st->lengths[ci] = (uint8_t)len;
    }
    if (lane == 0) {
        st->alphabet = (uint32_t)n;
        st->run_pairs = pairs;
        st->error = s_err;
    }
}


#endif /* HYDK_LF_HUFFMAN_H_ */
