/*
 * lf_coder.hip — the LF-group coder on the GPU.
 *
 * What it replaces (file:line relative to /root/reference/src/libhydrium/): the LF-coefficient
 * sub-stream of write_lf_group (encoder.c:560-596): clamped-gradient prediction of the LF ints,
 * hyd_entropy_send_symbol with LZ77 used as run-length coding (entropy.c:473-524), the
 * depth-limited Huffman construction with its selection order (entropy.c:577-662), canonical code
 * assignment (entropy.c:664-707) and the symbol write-out (entropy.c:1003-1021).  The constant
 * sub-streams around it (MA tree, HF metadata) and the code-length header stay on the host
 * (csrc/host/frame.c), which receives the code lengths and the packed symbol bits from here.
 *
 * The stream is one value sequence per LF group: channels Y, X, B, raster inside a channel,
 * n = 3 * vbw * vbh <= 196608 values.  Everything is cut into windows that one 256-thread workgroup
 * handles, so that an LF group is hundreds of small, short-lived workgroups that fit beside whatever
 * else the GPU is running (a 1024-thread workgroup per LF group, as in round 1, had to wait for a
 * compute unit with sixteen free wave slots and then held half of its registers for a millisecond:
 * 16 % of the pipelined frame rate for 1.5 % of its instructions):
 *
 *   k_lf_tokens   one workgroup per 896 values: residuals -> run structure -> one 8-byte record per
 *                 value + token histogram.  A maximal run of equal values is cut into chunks of 128:
 *                 the chunk's first value is a literal; the r <= 127 repeats behind it become one (run
 *                 token r - 3, distance) pair when r > 3 and r literals otherwise.  What a position
 *                 emits therefore depends only on its offset in its run and on at most 127 values
 *                 ahead: the workgroup looks at a window of 1024 values and finds the run its first
 *                 value continues by a (parallel) backward scan.
 *   k_lf_huffman  one wavefront per LF group: the reference's O(n^2) selection loop with the two
 *                 smallest candidates found by a wave-wide minimum over a total order that reproduces
 *                 its comparator and slot-visiting order; subtree depth recursion replaced by subtree
 *                 heights and a parent walk.  Runs in a compact slot space (only the slots the
 *                 16509-entry alphabet can ever touch) with all candidates held in registers: no
 *                 LDS traffic or barrier inside the merge loop.
 *   k_lf_offsets  one workgroup per LF group: bits of each window (its token histogram times the code
 *                 lengths, plus its residue bits), their exclusive prefix sum; clears the words two
 *                 windows share.
 *   k_lf_pack     one workgroup per window: per-value bit strings (<= 59 bits) ORed into an LDS
 *                 window at the window's bit offset, whole words stored, the two shared words ORed into
 *                 memory; LSB-first like HYDBitWriter (bitwriter.c:110-124).
 *
 * The work is small (<= 196608 values per 4.2 Mpx LF group); it exists so that a frame's sections are
 * complete on the device and the host's per-frame serial work disappears.
 */
#include <hip/hip_runtime.h>

#include <stdint.h>

#include "hydk_common.h"
/* HYDK_LF_WPB: windows a workgroup of k_lf_tokens / k_lf_pack takes, one after another (1: 220 workgroups per LF group and kernel) */
#ifndef HYDK_LF_WPB
#define HYDK_LF_WPB 1
#endif
/* HYDK_LF_PROBE (timing only, wrong LF streams): 1 no s_setprio in the LF kernels; 2 k_lf_tokens leaves no histograms (no global
 * atomics); 4 k_lf_tokens stores no records; 8 k_lf_tokens' workgroups return at once (what do 7 040 workgroups cost by existing?) */
#ifndef HYDK_LF_PROBE
#define HYDK_LF_PROBE 0
#endif

namespace {

#include "lf_huffman.h"

constexpr int kPlane = HYDK_DC_PITCH * HYDK_DC_PITCH;

/* record: bits 0-31 value, bit 32 "emit a literal", bits 33-39 run length r (0: no run pair) */
#define LF_REC(v, lit, r) ((unsigned long long)(v) | ((unsigned long long)((uint32_t)(lit) | ((uint32_t)(r) << 1)) << 32))

struct LfShape {
    int vbw, blocks, n;
    uint32_t vbw_magic; /* ceil(2^32 / vbw): rem / vbw = mulhi(rem, magic) for rem < 2^16 (vbw = 1: handled apart) */
};

__device__ __forceinline__ LfShape lf_shape(const HydkLfJob &job) {
    LfShape sh;
    sh.vbw = (job.width + 7) >> 3;
    sh.blocks = sh.vbw * ((job.height + 7) >> 3);
    sh.n = 3 * sh.blocks;
    sh.vbw_magic = sh.vbw > 1 ? (uint32_t)(0xFFFFFFFFu / (uint32_t)sh.vbw) + 1u : 0u;
    return sh;
}

/* rem / vbw for 0 <= rem < 65536 without the 30-instruction integer division: the multiply-high by
 * ceil(2^32 / vbw) overshoots rem / vbw by less than 2^-16, never across an integer (vbw <= 256) */
__device__ __forceinline__ int lf_row_of(const LfShape &sh, int rem) {
    return sh.vbw > 1 ? (int)__umulhi((uint32_t)rem, sh.vbw_magic) : rem;
}

/* hybrid-uint config (split 7, msb 1, lsb 1), entropy.c:427-444 */
__device__ __forceinline__ void lf_hybrid(uint32_t v, uint32_t &token, uint32_t &nbits, uint32_t &residue) {
    if (v < 128u) {
        token = v;
        nbits = 0;
        residue = 0;
        return;
    }
    const uint32_t nb = (31u - (uint32_t)__clz(v)) - 2u;
    residue = (v >> 1) & ((1u << nb) - 1u);
    token = 128u + (((nb - 5u) << 2) | (((v >> (nb + 1u)) & 1u) << 1) | (v & 1u));
    nbits = nb;
}

#define LF_DPP_KEEP(v, ctrl, rmask) ((uint32_t)__builtin_amdgcn_update_dpp((int)(v), (int)(v), ctrl, rmask, 0xF, false))
#define LF_DPP_ZERO(v, ctrl, rmask) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, rmask, 0xF, false))

/* inclusive scans over the wavefront: row_shr 1/2/4/8 inside each row of 16, then row_bcast 15 / 31 */
__device__ __forceinline__ int wave_incl_max(int v) {
    int t;
    t = (int)LF_DPP_KEEP(v, 0x111, 0xF);
    v = v > t ? v : t;
    t = (int)LF_DPP_KEEP(v, 0x112, 0xF);
    v = v > t ? v : t;
    t = (int)LF_DPP_KEEP(v, 0x114, 0xF);
    v = v > t ? v : t;
    t = (int)LF_DPP_KEEP(v, 0x118, 0xF);
    v = v > t ? v : t;
    t = (int)LF_DPP_KEEP(v, 0x142, 0xA);
    v = v > t ? v : t;
    t = (int)LF_DPP_KEEP(v, 0x143, 0xC);
    v = v > t ? v : t;
    return v;
}

__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t v) {
    v += LF_DPP_ZERO(v, 0x111, 0xF);
    v += LF_DPP_ZERO(v, 0x112, 0xF);
    v += LF_DPP_ZERO(v, 0x114, 0xF);
    v += LF_DPP_ZERO(v, 0x118, 0xF);
    v += LF_DPP_ZERO(v, 0x142, 0xA);
    v += LF_DPP_ZERO(v, 0x143, 0xC);
    return v;
}

/* ==========================================================================================
 * tokens: residuals -> run structure -> records + token histogram, one window per workgroup.
 * Thread t owns window positions 4t .. 4t+3.
 * ======================================================================================== */
struct LfTokenScratch {
    int wtot[kLfWaves];       /* per wave: last run head inside it (absolute index) or -1 */
    uint32_t lastv[kLfWaves]; /* per wave: its last value (the next wave's lane 0 compares against it) */
    int head[kLfWaves];       /* backward scan: nearest run head each wave found, or -1 */
    uint32_t first;           /* the window's first value */
};

/* value of the stream at plane c, block (y, x): pack_signed(lf - clamped_gradient(w, n, nw)),
 * encoder.c:574-594, in 32-bit wrap-around arithmetic (as the reference's int32 code behaves on the
 * targets it runs on) */
__device__ __forceinline__ uint32_t lf_predict(int32_t cur, int32_t w, int32_t n, int32_t nw) {
    const int32_t lo = w < n ? w : n, hi = w < n ? n : w;
    int32_t pred = (int32_t)((uint32_t)w + (uint32_t)n - (uint32_t)nw);
    pred = pred < lo ? lo : pred > hi ? hi : pred;
    const uint32_t d = (uint32_t)cur - (uint32_t)pred;
    return (d << 1) ^ (0u - (d >> 31));
}

__device__ __forceinline__ uint32_t lf_residual_at(const int32_t *dc, int c, int y, int x) {
    const int idx = c * kPlane + y * HYDK_DC_PITCH + x;
    const int iw = x ? idx - 1 : y ? idx - HYDK_DC_PITCH : idx;
    const int in = y ? idx - HYDK_DC_PITCH : iw;
    const int inw = x && y ? idx - HYDK_DC_PITCH - 1 : iw;
    const int32_t cur = dc[idx];
    int32_t w = dc[iw], n = dc[in], nw = dc[inw];
    if (!(x | y)) /* top-left block: all three neighbours count as 0 */
        w = n = nw = 0;
    return lf_predict(cur, w, n, nw);
}

/* the stream's value at position i (0 <= i < n) */
__device__ __forceinline__ uint32_t lf_value_at(const int32_t *dc, const LfShape &sh, int i) {
    const int visit = (i >= sh.blocks) + (i >= 2 * sh.blocks);
    const int rem = i - visit * sh.blocks;
    const int y = lf_row_of(sh, rem), x = rem - y * sh.vbw;
    return lf_residual_at(dc, visit < 2 ? 1 - visit : 2, y, x);
}

/* the thread's four consecutive values from stream position i0 on.  The usual case — all four in one
 * block row — shares its neighbours: ten loads instead of sixteen, no per-value address arithmetic. */
__device__ __forceinline__ void lf_fetch(const int32_t *dc, const LfShape &sh, int i0, uint32_t (&v)[4]) {
#pragma unroll
    for (int j = 0; j < 4; j++)
        v[j] = 0;
    if (i0 >= sh.n)
        return;
    int visit = (i0 >= sh.blocks) + (i0 >= 2 * sh.blocks);
    const int rem = i0 - visit * sh.blocks;
    int y = lf_row_of(sh, rem), x = rem - y * sh.vbw;
    if (x + 3 < sh.vbw) { /* same row, hence same channel and inside the stream */
        const int c = visit < 2 ? 1 - visit : 2; /* Y, X, B */
        const int32_t *row = dc + c * kPlane + y * HYDK_DC_PITCH + x;
        const int32_t *up = y ? row - HYDK_DC_PITCH : row;
        const int32_t c0 = row[0], c1 = row[1], c2 = row[2], c3 = row[3];
        const int32_t u0 = up[0], u1 = up[1], u2 = up[2], u3 = up[3];
        const int32_t left = row[x ? -1 : 0], upleft = up[x ? -1 : 0];
        const int32_t w0 = x ? left : y ? u0 : 0;
        v[0] = lf_predict(c0, w0, y ? u0 : w0, y && x ? upleft : w0);
        v[1] = lf_predict(c1, c0, y ? u1 : c0, y ? u0 : c0);
        v[2] = lf_predict(c2, c1, y ? u2 : c1, y ? u1 : c1);
        v[3] = lf_predict(c3, c2, y ? u3 : c2, y ? u2 : c2);
        return;
    }
    const int vbh = sh.blocks / sh.vbw;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (visit < 3)
            v[j] = lf_residual_at(dc, visit < 2 ? 1 - visit : 2, y, x);
        if (++x == sh.vbw) {
            x = 0;
            if (++y == vbh) {
                y = 0;
                visit++;
            }
        }
    }
}

/* Start of the run that position `from` (>= 0) belongs to: the largest h <= from with h == 0 or
 * value[h] != value[h - 1].  All threads of the workgroup scan backwards together, 256 positions per
 * step; photographic content stops in the first step, a flat plane walks all of it (768 steps at most). */
__device__ __forceinline__ int lf_run_start(const int32_t *dc, const LfShape &sh, int from, LfTokenScratch &S) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int top = from; top >= 0; top -= kLfThreads) {
        const int p = top - tid;
        int head = -1;
        if (p >= 0)
            head = (p == 0 || lf_value_at(dc, sh, p) != lf_value_at(dc, sh, p - 1)) ? p : -1;
        head = wave_incl_max(head);
        if (lane == 63)
            S.head[wave] = head;
        __syncthreads();
        int best = -1;
#pragma unroll
        for (int w = 0; w < kLfWaves; w++)
            best = best > S.head[w] ? best : S.head[w];
        __syncthreads();
        if (best >= 0)
            return best;
    }
    return 0;
}

/* returns the residue bits of the literals this thread sent */
__device__ __forceinline__ uint32_t lf_tokens_window(const HydkLfJob &job, const LfShape &sh, unsigned long long *__restrict__ recs,
                                                     int tb, int *s_rs /* [kScanSpan] */, uint32_t *s_hist, LfTokenScratch &S) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int32_t *dc = job.dc;
    const int q0 = tid * 4;

    for (int i = tid; i < HYDK_LF_CODES; i += kLfThreads)
        s_hist[i] = 0;
    uint32_t v[4];
    lf_fetch(dc, sh, tb + q0, v);
    /* what lies in front of the window: the value at tb - 1 and — only if the window's first value continues its run,
     * which photographic content hardly ever does — the start of that run (a backward scan by the whole workgroup:
     * until round 3 every window paid for it, a third of this kernel's instructions) */
    const uint32_t tailv = tb > 0 ? lf_value_at(dc, sh, tb - 1) : 0u;
    if (lane == 63)
        S.lastv[wave] = v[3];
    if (tid == 0)
        S.first = v[0];
    __syncthreads();
    const bool continues = tb > 0 && tb < sh.n && S.first == tailv; /* the same for every thread */
    const int carry = continues ? lf_run_start(dc, sh, tb - 1, S) : 0;

    /* start of the run each position belongs to (absolute index): max-scan of run heads */
    uint32_t prev = LF_DPP_KEEP(v[3], 0x138, 0xF); /* wave_shr:1 — the previous lane's last value */
    if (lane == 0)
        prev = wave ? S.lastv[wave - 1] : tailv;
    int rs[4];
    int last = -1;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int i = tb + q0 + j;
        last = (i == 0 || i >= sh.n || v[j] != prev) ? i : last;
        rs[j] = last;
        prev = v[j];
    }
    const int inc = wave_incl_max(last);
    if (lane == 63)
        S.wtot[wave] = inc;
    __syncthreads();
    int pre = carry;
    for (int w = 0; w < wave; w++) {
        const int t = S.wtot[w];
        pre = pre > t ? pre : t;
    }
    {
        int excl = (int)LF_DPP_KEEP(inc, 0x138, 0xF);
        excl = lane ? excl : -1;
        pre = pre > excl ? pre : excl;
    }
#pragma unroll
    for (int j = 0; j < 4; j++)
        rs[j] = rs[j] < 0 ? pre : rs[j];
    *(int4 *)&s_rs[q0] = make_int4(rs[0], rs[1], rs[2], rs[3]);
    __syncthreads();

    /* what each position sends, without divergent control flow: offset c inside the run's current
     * 128-chunk; "same4" = the chunk has a fifth value, i.e. more than 3 repeats behind its first */
    const int last_q = sh.n - 1 - tb; /* window-relative index of the stream's last value */
    uint32_t lit[4], r[4];
    bool need[4], any_need = false;
    int s4v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int q = q0 + j;
        const bool valid = q < kEmitSpan && q <= last_q;
        const int c = (tb + q - rs[j]) & 127;
        const int s4 = q - c + 4;
        const bool same4 = c <= 3 && s4 <= last_q && s_rs[s4 < kScanSpan ? s4 : kScanSpan - 1] == rs[j];
        lit[j] = valid && (c == 0 || (c <= 3 && !same4));
        need[j] = valid && c == 0 && same4;
        r[j] = 0;
        s4v[j] = s4;
        any_need |= need[j];
    }
    if (__any(any_need)) { /* some chunk head has a run behind it: how long, up to 127 (7 halvings) */
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int q = q0 + j;
            int lo = need[j] ? s4v[j] : 0, hi = q + 127 < last_q ? q + 127 : last_q;
            hi = need[j] ? hi : 0;
#pragma unroll
            for (int it = 0; it < 7; it++) {
                const int mid = (lo + hi + 1) >> 1;
                const bool in_run = s_rs[mid] == rs[j];
                lo = in_run ? mid : lo;
                hi = in_run ? hi : mid - 1;
            }
            r[j] = need[j] ? (uint32_t)(lo - q) : 0u;
        }
    }
    uint32_t residue_bits = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (lit[j]) {
            uint32_t token, nb, res;
            lf_hybrid(v[j], token, nb, res);
            atomicAdd(&s_hist[token], 1u);
            residue_bits += nb;
        }
        if (r[j])
            atomicAdd(&s_hist[256u + r[j] - 3u], 1u);
    }
    if (!(HYDK_LF_PROBE & 4) && q0 < kEmitSpan && q0 <= last_q) { /* a thread's four records are 32 contiguous bytes; the tail of the stream is padded, never read */
        ulonglong2 *dst = (ulonglong2 *)(recs + tb + q0);
        dst[0] = make_ulonglong2(q0 + 0 <= last_q ? LF_REC(v[0], lit[0], r[0]) : 0ull, q0 + 1 <= last_q ? LF_REC(v[1], lit[1], r[1]) : 0ull);
        dst[1] = make_ulonglong2(q0 + 2 <= last_q ? LF_REC(v[2], lit[2], r[2]) : 0ull, q0 + 3 <= last_q ? LF_REC(v[3], lit[3], r[3]) : 0ull);
    }
    __syncthreads();
    return residue_bits;
}

/* ==========================================================================================
 * count / offsets / pack: per-value bit strings -> bit offsets -> bits.  A window is 1024 values
 * (four per thread); a value takes at most 59 bits.
 * ======================================================================================== */
constexpr int kPackWords = (31 + kEmitSpan * 59) / 32 + 2;

/* the bit string (value, length) each of the thread's four records sends under the LF group's code;
 * returns their total length */
__device__ __forceinline__ uint32_t lf_window_strings(const LfShape &sh, const unsigned long long *__restrict__ recs, int tb,
                                                      const uint32_t *s_code, unsigned long long (&val)[4], uint32_t (&len)[4]) {
    const int i0 = tb + (int)threadIdx.x * 4;
    unsigned long long rec[4] = {0, 0, 0, 0};
    const bool mine_window = (int)threadIdx.x * 4 < kEmitSpan; /* the last 32 threads have no values of this window */
    if (mine_window && i0 < sh.n) { /* records are stored in padded groups of four */
        const ulonglong2 a = ((const ulonglong2 *)(recs + i0))[0], b = ((const ulonglong2 *)(recs + i0))[1];
        rec[0] = a.x;
        rec[1] = a.y;
        rec[2] = b.x;
        rec[3] = b.y;
    }
    uint32_t mine = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        val[j] = 0;
        len[j] = 0;
        if (mine_window && i0 + j < sh.n) {
            const uint32_t v = (uint32_t)rec[j], lit = (uint32_t)(rec[j] >> 32) & 1u, r = (uint32_t)(rec[j] >> 33) & 127u;
            if (lit) {
                uint32_t token, nb, res;
                lf_hybrid(v, token, nb, res);
                const uint32_t e = s_code[token];
                val[j] = (e & 0xFFFFu) | ((unsigned long long)res << (e >> 16));
                len[j] = (e >> 16) + nb;
            }
            if (r) {
                const uint32_t e = s_code[256u + r - 3u];
                val[j] |= (unsigned long long)(e & 0xFFFFu) << len[j];
                len[j] += e >> 16;
            }
        }
        mine += len[j];
    }
    return mine;
}

/* ==========================================================================================
 * The kernels.  All workgroups are 256 threads and live for microseconds.
 * ======================================================================================== */
/* grid = (windows of 896 values, LF groups); hist_all must be zero on entry */
__global__ __launch_bounds__(kLfThreads) void k_lf_tokens(const HydkLfJob *__restrict__ jobs,
                                                          unsigned long long *__restrict__ recs_all,
                                                          uint32_t *__restrict__ hist_all, LfWork *__restrict__ work) {
    if (!(HYDK_LF_PROBE & 1))
        __builtin_amdgcn_s_setprio(3); /* late work of a frame whose stream holds nothing else: see kernels.hip HYDK_URGENT */
    const int slot = blockIdx.y, tid = threadIdx.x;
    const HydkLfJob &job = jobs[slot];
    const LfShape sh = lf_shape(job);
    __shared__ __attribute__((aligned(16))) int s_rs[kScanSpan];
    __shared__ uint32_t s_hist[HYDK_LF_CODES];
    __shared__ LfTokenScratch s_tok;
    __shared__ uint32_t s_rbits;
    if (HYDK_LF_PROBE & 8)
        return;
    /* HYDK_LF_WPB windows per workgroup, one after another (1: a workgroup per window) */
    for (int win = (int)blockIdx.x * HYDK_LF_WPB; win < ((int)blockIdx.x + 1) * HYDK_LF_WPB && win < kMaxWindows; win++) {
        const int tb = win * kEmitSpan;
        if (tb >= sh.n)
            return;
        if (tid == 0)
            s_rbits = 0;
        uint32_t rb = lf_tokens_window(job, sh, recs_all + (size_t)slot * HYDK_LF_SYMBOLS, tb, s_rs, s_hist, s_tok);
        rb = wave_incl_sum(rb);
        if ((tid & 63) == 63 && rb)
            atomicAdd(&s_rbits, rb);
        __syncthreads();
        LfWork &w = work[slot];
        for (int i = tid; i < HYDK_LF_CODES && !(HYDK_LF_PROBE & 2); i += kLfThreads) {
            const uint32_t c = s_hist[i];
            w.win_hist[win][i] = (uint16_t)c; /* at most 896 + 7 per window */
            if (c)
                atomicAdd(&hist_all[(size_t)slot * HYDK_LF_CODES + i], c);
        }
        if (tid == 0)
            w.win_residue_bits[win] = s_rbits;
        if (HYDK_LF_WPB > 1)
            __syncthreads(); /* the next window reuses the scratch */
    }
}

/* grid = LF groups, block = 64: code lengths + canonical codes of each LF group's histogram */
__global__ __launch_bounds__(64) void k_lf_codes(const uint32_t *__restrict__ hist_all, HydkLfStream *__restrict__ streams,
                                                 LfWork *__restrict__ work) {
    if (!(HYDK_LF_PROBE & 1))
        __builtin_amdgcn_s_setprio(3); /* late work of a frame whose stream holds nothing else: see kernels.hip HYDK_URGENT */
    __shared__ LfHuffScratch s_huff;
    const int slot = blockIdx.x;
    lf_huffman_wave(hist_all + (size_t)slot * HYDK_LF_CODES, work[slot].codes, streams + slot, s_huff, (int)threadIdx.x);
}

/* grid = LF groups, block = 256: bits of each window = sum over tokens of (count x code length) + its
 * residue bits; where each window's bits start; the words two windows share are cleared */
__global__ __launch_bounds__(kLfThreads) void k_lf_offsets(const HydkLfJob *__restrict__ jobs, LfWork *__restrict__ work,
                                                           HydkLfStream *__restrict__ streams, uint32_t *__restrict__ bits_all) {
    if (!(HYDK_LF_PROBE & 1))
        __builtin_amdgcn_s_setprio(3); /* late work of a frame whose stream holds nothing else: see kernels.hip HYDK_URGENT */
    const int slot = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const LfShape sh = lf_shape(jobs[slot]);
    const int windows = (sh.n + kEmitSpan - 1) / kEmitSpan; /* <= 220 < 256 */
    LfWork &w = work[slot];
    __shared__ uint32_t s_win[kMaxWindows];
    __shared__ uint32_t s_wsum[kLfWaves];
    /* a wave per window: lanes over the 384 tokens (6 each) */
    uint32_t len6[6];
#pragma unroll
    for (int j = 0; j < 6; j++)
        len6[j] = w.codes[j * 64 + lane] >> 16;
    for (int win = wave; win < windows; win += kLfWaves) {
        uint32_t bits = 0;
#pragma unroll
        for (int j = 0; j < 6; j++)
            bits += (uint32_t)w.win_hist[win][j * 64 + lane] * len6[j];
        bits = wave_incl_sum(bits);
        if (lane == 63)
            s_win[win] = bits + w.win_residue_bits[win];
    }
    __syncthreads();
    const uint32_t mine = tid < windows ? s_win[tid] : 0u;
    const uint32_t inc = wave_incl_sum(mine);
    if (lane == 63)
        s_wsum[wave] = inc;
    __syncthreads();
    uint32_t off = inc - mine, total = 0;
    for (int k = 0; k < kLfWaves; k++) {
        if (k < wave)
            off += s_wsum[k];
        total += s_wsum[k];
    }
    uint32_t *out = bits_all + (size_t)slot * HYDK_LF_BITWORDS;
    if (tid < windows) {
        w.win_off[tid] = off;
        /* the window's first word may hold the end of the window before, its last the start of the one
         * after: k_lf_pack ORs into exactly these two */
        if (mine) {
            out[off >> 5] = 0;
            out[(off + mine - 1u) >> 5] = 0;
        }
    }
    if (tid == 0)
        streams[slot].bit_count = total;
}

/* grid = (windows of 896 values, LF groups) */
__global__ __launch_bounds__(kLfThreads) void k_lf_pack(const HydkLfJob *__restrict__ jobs,
                                                        const unsigned long long *__restrict__ recs_all,
                                                        const LfWork *__restrict__ work, uint32_t *__restrict__ bits_all) {
    if (!(HYDK_LF_PROBE & 1))
        __builtin_amdgcn_s_setprio(3); /* late work of a frame whose stream holds nothing else: see kernels.hip HYDK_URGENT */
    const int slot = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const LfShape sh = lf_shape(jobs[slot]);
    __shared__ uint32_t s_bits[kPackWords];
    __shared__ uint32_t s_code[HYDK_LF_CODES];
    __shared__ uint32_t s_wsum[kLfWaves];
    if ((int)blockIdx.x * HYDK_LF_WPB * kEmitSpan >= sh.n)
        return;
    for (int i = tid; i < HYDK_LF_CODES; i += kLfThreads)
        s_code[i] = work[slot].codes[i];
  for (int win = (int)blockIdx.x * HYDK_LF_WPB; win < ((int)blockIdx.x + 1) * HYDK_LF_WPB && win < kMaxWindows; win++) {
    const int tb = win * kEmitSpan;
    if (tb >= sh.n)
        return;
    if (win != (int)blockIdx.x * HYDK_LF_WPB)
        __syncthreads(); /* the window before has left s_bits and s_wsum */
    for (int w = tid; w < kPackWords; w += kLfThreads)
        s_bits[w] = 0;
    __syncthreads();
    unsigned long long val[4];
    uint32_t len[4];
    const uint32_t mine = lf_window_strings(sh, recs_all + (size_t)slot * HYDK_LF_SYMBOLS, tb, s_code, val, len);
    const uint32_t inc = wave_incl_sum(mine);
    if (lane == 63)
        s_wsum[wave] = inc;
    __syncthreads();
    const uint32_t gbits = work[slot].win_off[win]; /* bits in front of this window */
    uint32_t pos = (gbits & 31u) + inc - mine, total = 0;
    for (int w = 0; w < kLfWaves; w++) {
        const uint32_t t = s_wsum[w];
        if (w < wave)
            pos += t;
        total += t;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (!len[j])
            continue;
        const uint32_t w = pos >> 5, shl = pos & 31u;
        const unsigned long long lo = val[j] << shl;
        const uint32_t hi = shl ? (uint32_t)(val[j] >> (64u - shl)) : 0u;
        if ((uint32_t)lo)
            atomicOr(&s_bits[w], (uint32_t)lo);
        if ((uint32_t)(lo >> 32))
            atomicOr(&s_bits[w + 1], (uint32_t)(lo >> 32));
        if (hi)
            atomicOr(&s_bits[w + 2], hi);
        pos += len[j];
    }
    __syncthreads();
    /* whole words are stored; the first and the last word of the span may be shared with the neighbouring
     * windows (k_lf_offsets cleared them): those are ORed into memory */
    if (!total)
        continue;
    const uint32_t end = (gbits & 31u) + total, last = (end - 1u) >> 5;
    uint32_t *dst = bits_all + (size_t)slot * HYDK_LF_BITWORDS + (gbits >> 5);
    for (uint32_t w = tid; w <= last; w += kLfThreads) {
        if (w == 0 || w == last)
            atomicOr(&dst[w], s_bits[w]);
        else
            dst[w] = s_bits[w];
    }
  }
}

/* The LF groups' symbol data, 4-byte aligned, back to back in slot order: one copy (or one
 * all-gather) moves a frame's LF streams.  grid = LF groups, block = 256. */
__global__ __launch_bounds__(256) void k_lf_gather(HydkLfStream *__restrict__ streams, const uint32_t *__restrict__ bits_all,
                                                   uint32_t *__restrict__ packed, unsigned long long *__restrict__ total,
                                                   int num_slots) {
    const int slot = blockIdx.x, tid = threadIdx.x;
    __shared__ uint32_t s_off;
    if (tid < 64) {
        uint32_t mine = 0;
        for (int s = tid; s < slot; s += 64)
            mine += (streams[s].bit_count + 31u) >> 5;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1)
            mine += __shfl_xor(mine, d);
        if (tid == 0)
            s_off = mine;
    }
    __syncthreads();
    const uint32_t off = s_off, words = (streams[slot].bit_count + 31u) >> 5;
    const uint32_t *src = bits_all + (size_t)slot * HYDK_LF_BITWORDS;
    for (uint32_t w = tid; w < words; w += 256)
        packed[off + w] = src[w];
    if (tid == 0) {
        streams[slot].offset = off * 4u;
        if (slot == num_slots - 1)
            *total = (unsigned long long)(off + words) * 4ull;
    }
}

/* unit-test entry: the code construction alone, on a histogram in global memory */
__global__ __launch_bounds__(64) void k_lf_huffman(const uint32_t *__restrict__ hist, HydkLfStream *__restrict__ stream_out,
                                                   uint32_t *__restrict__ codes) {
    __shared__ LfHuffScratch s_huff;
    lf_huffman_wave(hist, codes, stream_out, s_huff, (int)threadIdx.x);
}

} // namespace

namespace hydk {

/* bytes of scratch per LF group (device_api.hip allocates it) */
size_t lf_work_bytes() { return sizeof(LfWork); }

/* The LF coder for `num_slots` LF groups in three steps (every pointer addresses the first of the LF
 * groups): tokens; code construction; offsets + pack.  The middle step may instead ride in the entropy
 * stage's launch (kernels.hip k_rans_lanes), which is why it can be left out here. */
hipError_t launch_lf_front(const HydkLfJob *d_jobs, unsigned long long *recs, uint32_t *hist, void *work, int num_slots,
                           hipStream_t stream) {
    /* hist is zero on entry: it lives in the arena k_frame_begin clears once per frame */
    hipLaunchKernelGGL(k_lf_tokens, dim3((kMaxWindows + HYDK_LF_WPB - 1) / HYDK_LF_WPB, num_slots), dim3(kLfThreads), 0, stream, d_jobs, recs, hist, (LfWork *)work);
    return hipGetLastError();
}

hipError_t launch_lf_codes(const uint32_t *hist, HydkLfStream *streams, void *work, int num_slots, hipStream_t stream) {
    hipLaunchKernelGGL(k_lf_codes, dim3(num_slots), dim3(64), 0, stream, hist, streams, (LfWork *)work);
    return hipGetLastError();
}

hipError_t launch_lf_back(const HydkLfJob *d_jobs, const unsigned long long *recs, HydkLfStream *streams, uint32_t *bits,
                          void *work, int num_slots, hipStream_t stream) {
    LfWork *w = (LfWork *)work;
    hipLaunchKernelGGL(k_lf_offsets, dim3(num_slots), dim3(kLfThreads), 0, stream, d_jobs, w, streams, bits);
    hipLaunchKernelGGL(k_lf_pack, dim3((kMaxWindows + HYDK_LF_WPB - 1) / HYDK_LF_WPB, num_slots), dim3(kLfThreads), 0, stream, d_jobs, recs, w, bits);
    return hipGetLastError();
}

hipError_t launch_lf_coder(const HydkLfJob *d_jobs, unsigned long long *recs, uint32_t *hist, HydkLfStream *streams,
                           uint32_t *bits, void *work, int num_slots, hipStream_t stream) {
    hipError_t e = launch_lf_front(d_jobs, recs, hist, work, num_slots, stream);
    if (e == hipSuccess)
        e = launch_lf_codes(hist, streams, work, num_slots, stream);
    if (e == hipSuccess)
        e = launch_lf_back(d_jobs, recs, streams, bits, work, num_slots, stream);
    return e;
}

/* once per frame, over all of its LF groups */
hipError_t launch_lf_gather(HydkLfStream *streams, const uint32_t *bits, uint32_t *packed, unsigned long long *total,
                            int num_slots, hipStream_t stream) {
    hipLaunchKernelGGL(k_lf_gather, dim3(num_slots), dim3(256), 0, stream, streams, bits, packed, total, num_slots);
    return hipGetLastError();
}

/* debug / unit-test entry: code lengths and codes for one caller-supplied histogram (device pointers) */
hipError_t launch_lf_huffman_only(const uint32_t *hist, HydkLfStream *stream_out, uint32_t *codes, hipStream_t stream) {
    hipLaunchKernelGGL(k_lf_huffman, dim3(1), dim3(64), 0, stream, hist, stream_out, codes);
    return hipGetLastError();
}

} // namespace hydk
