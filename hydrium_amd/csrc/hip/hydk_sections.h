/*
 * hydk_sections.h — the data-dependent bit fields of a frame's small sections, written so that the
 * same source runs on the GPU (csrc/hip/assemble.hip, one lane or thread per field) and on the host
 * (the CPU-only tests compare it with csrc/host/prefix.c / frame.c field by field).
 *
 * What is restated here (file:line relative to /root/reference/src/libhydrium/):
 *   hydk_put_ans_distribution   one 12-bit ANS histogram (entropy.c:303-369)
 *   hydk_small_code_lengths     the depth-limited Huffman selection (entropy.c:577-662) for the 18-symbol
 *                               code of code lengths
 *   hydk_lf_prefix_codes        alphabet sizes + one prefix code per cluster of the LF-coefficient stream
 *                               (entropy.c:835-927 with the code-length coding of entropy.c:709-805)
 *   hydk_toc_entry              a TOC size as U32(0+u10, 1024+u14, 17408+u22, 4211712+u30) (encoder.c:117-120)
 *
 * A sink is an array of zeroed 32-bit words that bits are ORed into, LSB first (bitwriter.c:110-124).
 */
#ifndef HYDK_SECTIONS_H_
#define HYDK_SECTIONS_H_

#include <stdint.h>

#if defined(__HIPCC__)
#define HYDK_HD __host__ __device__ inline
#else
#define HYDK_HD static inline
#endif

#ifndef HYDK_LF_CODES
#define HYDK_LF_CODES 384
#endif
#ifndef HYDK_LF_RUN_BASE
#define HYDK_LF_RUN_BASE 16384
#endif

typedef struct HydkSink {
    uint32_t *w;       /* zeroed words; NULL: count only */
    uint64_t pos;      /* bits written so far (= where the next bit goes) */
    uint64_t cap_bits; /* bits the array holds */
    int overflow;      /* a put went past cap_bits (nothing was written for it) */
    int shared;        /* device only: other threads OR into the same words */
} HydkSink;

HYDK_HD void hks_or(HydkSink *s, uint64_t word, uint32_t v) {
    if (!v)
        return;
#if defined(__HIP_DEVICE_COMPILE__)
    if (s->shared)
        atomicOr(&s->w[word], v);
    else
        s->w[word] |= v;
#else
    s->w[word] |= v;
#endif
}

/* the low n bits of v (n <= 32) */
HYDK_HD void hks_put(HydkSink *s, uint32_t v, uint32_t n) {
    if (!n)
        return;
    if (n < 32)
        v &= (1u << n) - 1u;
    if (s->pos + n > s->cap_bits) {
        s->overflow = 1;
        s->pos += n;
        return;
    }
    if (s->w) {
        const uint64_t word = s->pos >> 5;
        const uint32_t sh = (uint32_t)(s->pos & 31u);
        hks_or(s, word, v << sh);
        if (sh + n > 32)
            hks_or(s, word + 1, v >> (32u - sh));
    }
    s->pos += n;
}

HYDK_HD void hks_put64(HydkSink *s, uint64_t v, uint32_t n) {
    if (n > 32) {
        hks_put(s, (uint32_t)v, 32);
        hks_put(s, (uint32_t)(v >> 32), n - 32);
    } else {
        hks_put(s, (uint32_t)v, n);
    }
}

HYDK_HD int hks_ilog2(uint32_t v) { /* v > 0 */
    int l = 0;
    while (v >>= 1)
        l++;
    return l;
}
HYDK_HD int hks_clog2(uint32_t v) { return v <= 1 ? 0 : hks_ilog2(v - 1) + 1; }

/* ---------------------------------------------------------------------------------------------
 * ANS histogram (entropy.c:303-369, 71-78)
 * ------------------------------------------------------------------------------------------- */
HYDK_HD void hks_ans_u8(HydkSink *s, uint32_t v) {
    hks_put(s, v != 0, 1);
    if (!v)
        return;
    const int l = hks_ilog2(v);
    hks_put(s, (uint32_t)l, 3);
    hks_put(s, v, (uint32_t)l);
}

HYDK_HD void hydk_put_ans_distribution(HydkSink *s, const uint32_t *freq, uint32_t alphabet) {
    /* (code, length) of log-count 0..13 (entropy.c:35-38) */
    const uint8_t log_code[14][2] = {{17, 5}, {11, 4}, {15, 4}, {3, 4}, {9, 4}, {7, 4}, {4, 3},
                                     {2, 3},  {5, 3},  {6, 3},  {0, 3}, {33, 6}, {1, 7}, {65, 7}};
    if (!alphabet) {
        hks_put(s, 1, 2); /* an unused cluster is sent as "always symbol 0" */
        hks_ans_u8(s, 0);
        return;
    }
    int first = -1, second = -1, seen = 0;
    for (uint32_t k = 0; k < alphabet; k++) {
        if (freq[k] == 4096) {
            hks_put(s, 1, 2);
            hks_ans_u8(s, k);
            return;
        }
        if (!freq[k])
            continue;
        if (++seen > 2)
            break;
        if (first < 0) {
            first = (int)k;
        } else if (freq[first] + freq[k] == 4096) {
            second = (int)k;
            break;
        }
    }
    if (first >= 0 && second >= 0) {
        hks_put(s, 3, 2);
        hks_ans_u8(s, (uint32_t)first);
        hks_ans_u8(s, (uint32_t)second);
        hks_put(s, freq[first], 12);
        return;
    }
    hks_put(s, 0, 2);
    hks_put(s, 7, 3);
    hks_put(s, 6, 3); /* together: shift = 13 */
    hks_ans_u8(s, alphabet - 3);
    uint32_t omit = 0;
    int omit_log = 0;
    for (uint32_t k = 0; k < alphabet; k++) {
        const int lc = freq[k] ? 1 + hks_ilog2(freq[k]) : 0;
        hks_put(s, log_code[lc][0], log_code[lc][1]);
        if (lc > omit_log) {
            omit_log = lc;
            omit = k;
        }
    }
    for (uint32_t k = 0; k < alphabet; k++) {
        const int lc = freq[k] ? 1 + hks_ilog2(freq[k]) : 0;
        if (k == omit || lc <= 1)
            continue;
        hks_put(s, freq[k], (uint32_t)(lc - 1));
    }
}

/* ---------------------------------------------------------------------------------------------
 * depth-limited code lengths for a small alphabet (entropy.c:577-662): the reference's selection
 * loop as it stands — round k settles slots 2k and 2k+1 as the children of node n + k; candidates are
 * the weighted slots in [2k, n + k) whose subtree may still grow; ties go to leaves before merged
 * nodes, leaves by symbol, and a merged node loses every comparison.
 * ------------------------------------------------------------------------------------------- */
#define HYDK_SMALL_N 18
typedef struct HydkSmallNode {
    uint32_t freq;
    int32_t token, depth, deepest, left, right;
} HydkSmallNode;

HYDK_HD int hks_node_before(const HydkSmallNode *a, const HydkSmallNode *b) {
    int32_t d;
    if (a->freq != b->freq)
        d = !b->freq ? -1 : !a->freq ? 1 : (int32_t)(a->freq - b->freq);
    else
        d = !b->token ? -1 : !a->token ? 1 : a->token - b->token;
    return d < 0;
}

/* returns 0, or a non-zero error; n <= HYDK_SMALL_N.  `nodes` [2 * HYDK_SMALL_N - 1] and `stack` [2 * HYDK_SMALL_N] are
 * the caller's workspace: on the device it sits in LDS — as private arrays they are indexed dynamically and so live in
 * scratch memory, where the ~1500 dependent accesses of an 18-symbol run cost 0.2 ms. */
HYDK_HD int hydk_small_code_lengths_ws(const uint32_t *freq, uint32_t *lengths, uint32_t n, int max_depth, HydkSmallNode *nodes,
                                       int32_t *stack) {
    uint32_t live_count = 0;
    for (uint32_t i = 0; i < 2 * n - 1; i++) {
        nodes[i].freq = i < n ? freq[i] : 0;
        nodes[i].token = i < n ? (int32_t)i + 1 : 0;
        nodes[i].depth = nodes[i].deepest = 0;
        nodes[i].left = nodes[i].right = -1;
        if (i < n)
            live_count += freq[i] != 0;
    }
    for (uint32_t i = 0; i < n; i++)
        lengths[i] = 0;
    if (!live_count)
        return 1;
    for (uint32_t k = 0; k + 1 < n; k++, live_count--) {
        const int32_t limit = max_depth - hks_clog2(live_count) + 1;
        /* the two best candidates, their sort keys kept in registers: hks_node_before(j, best) spelled out */
        int32_t first = -1, second = -1;
        uint32_t f1 = 0, f2 = 0;
        int32_t t1 = 0, t2 = 0;
        for (uint32_t j = 2 * k; j < n + k; j++) {
            const uint32_t fj = nodes[j].freq;
            if (!fj || nodes[j].deepest >= limit)
                continue;
            const int32_t tj = nodes[j].token;
            /* candidates all have a weight: smaller weight first; equal weights: a leaf before a merged node, leaves by
             * symbol, and a later merged node before an earlier one (hks_node_before with b->token == 0) */
            const int before1 = first < 0 || (fj != f1 ? fj < f1 : !t1 ? 1 : !tj ? 0 : tj < t1);
            if (before1) {
                second = first;
                f2 = f1;
                t2 = t1;
                first = (int32_t)j;
                f1 = fj;
                t1 = tj;
            } else if (second < 0 || (fj != f2 ? fj < f2 : !t2 ? 1 : !tj ? 0 : tj < t2)) {
                second = (int32_t)j;
                f2 = fj;
                t2 = tj;
            }
        }
        if (first < 0)
            return 2;
        if ((uint32_t)first != 2 * k) {
            const HydkSmallNode tmp = nodes[first];
            nodes[first] = nodes[2 * k];
            nodes[2 * k] = tmp;
        }
        if (second < 0)
            break; /* a single tree is left */
        if ((uint32_t)second == 2 * k)
            second = first;
        if ((uint32_t)second != 2 * k + 1) {
            const HydkSmallNode tmp = nodes[second];
            nodes[second] = nodes[2 * k + 1];
            nodes[2 * k + 1] = tmp;
        }
        HydkSmallNode *parent = &nodes[n + k];
        parent->freq = nodes[2 * k].freq + nodes[2 * k + 1].freq;
        parent->token = 0;
        parent->depth = parent->deepest = 0;
        parent->left = (int32_t)(2 * k);
        parent->right = (int32_t)(2 * k + 1);
        /* every node of the new subtree moves one level down, and with it the largest depth below it (`deepest`);
         * nothing outside the subtree changes.  The new root's own `deepest` is the larger of its children's. */
        int sp = 0;
        stack[sp++] = (int32_t)(n + k);
        while (sp) {
            const int32_t i = stack[--sp];
            nodes[i].depth++;
            nodes[i].deepest++;
            if (nodes[i].left >= 0)
                stack[sp++] = nodes[i].left;
            if (nodes[i].right >= 0)
                stack[sp++] = nodes[i].right;
        }
        {
            const int32_t dl = nodes[2 * k].deepest, dr = nodes[2 * k + 1].deepest;
            parent->deepest = dl > dr ? dl : dr; /* both >= 1, the depth the walk left the new root itself at */
        }
    }
    for (uint32_t j = 0; j < 2 * n - 1; j++)
        if (nodes[j].token)
            lengths[nodes[j].token - 1] = (uint32_t)nodes[j].depth;
    return 0;
}

HYDK_HD int hydk_small_code_lengths(const uint32_t *freq, uint32_t *lengths, uint32_t n, int max_depth) {
    HydkSmallNode nodes[2 * HYDK_SMALL_N - 1];
    int32_t stack[2 * HYDK_SMALL_N];
    return hydk_small_code_lengths_ws(freq, lengths, n, max_depth, nodes, stack);
}

/* canonical codes, bit-reversed for an LSB-first writer (entropy.c:664-707); returns 0 or an error */
HYDK_HD int hydk_small_codes(const uint32_t *lengths, uint32_t n, uint32_t *bits) {
    uint64_t next = 0;
    uint32_t longest = 0;
    for (uint32_t i = 0; i < n; i++)
        longest = lengths[i] > longest ? lengths[i] : longest;
    if (longest > 32)
        longest = 32;
    for (uint32_t len = 1; len <= longest; len++) {
        for (uint32_t i = 0; i < n; i++) {
            if (lengths[i] != len)
                continue;
            const uint32_t v = (uint32_t)(next >> (32 - len));
            uint32_t r = 0;
            for (uint32_t b = 0; b < len; b++)
                r |= ((v >> b) & 1u) << (len - 1 - b);
            bits[i] = r;
            next += (uint64_t)1 << (32 - len);
        }
    }
    return next && next != ((uint64_t)1 << 32) ? 3 : 0;
}

/* ---------------------------------------------------------------------------------------------
 * The LF-coefficient stream's header behind its fixed fields: alphabet sizes of the two clusters
 * (values, LZ77 distance) and one prefix code each (entropy.c:835-927).  lengths[] is indexed by
 * compact token — [0,256) literal tokens, [256,384) token 16384 + (i - 256) — as the LF coder leaves it.
 * ------------------------------------------------------------------------------------------- */
HYDK_HD uint32_t hks_lf_token(uint32_t compact) { return compact < 256 ? compact : HYDK_LF_RUN_BASE + (compact - 256); }

HYDK_HD void hks_zero_run(HydkSink *s, const uint32_t *l1_bits, const uint32_t *l1_len, uint32_t run) { /* entropy.c:709-728 */
    if (run >= 3) {
        uint32_t digits[8];
        int nd = 0;
        while (run > 10) {
            const uint32_t shorter = (run + 13) / 8;
            digits[nd++] = run - 8 * shorter + 16;
            run = shorter;
        }
        digits[nd++] = run;
        while (nd--) {
            hks_put(s, l1_bits[17], l1_len[17]);
            hks_put(s, digits[nd] - 3, 3);
        }
    } else {
        for (uint32_t k = 0; k < run; k++)
            hks_put(s, l1_bits[0], l1_len[0]);
    }
}

HYDK_HD int hks_complex_lengths(HydkSink *s, const uint8_t *lengths, uint32_t alphabet) { /* entropy.c:730-805 */
    const uint8_t order[18] = {1, 2, 3, 4, 0, 5, 17, 6, 16, 7, 8, 9, 10, 11, 12, 13, 14, 15}; /* entropy.c:42 */
    const uint8_t lol_bits[6] = {0, 7, 3, 2, 1, 15}, lol_len[6] = {2, 4, 3, 2, 2, 4};          /* entropy.c:44-46 */
    hks_put(s, 0, 2); /* hskip = 0 */
    uint32_t l1_freq[18], l1_len[18], l1_bits[18];
    for (int i = 0; i < 18; i++)
        l1_freq[i] = l1_len[i] = l1_bits[i] = 0;
    uint32_t prev = 0; /* token after the last coded one */
    for (uint32_t ci = 0; ci < HYDK_LF_CODES; ci++) {
        const uint32_t len = lengths[ci], tok = hks_lf_token(ci);
        if (!len || tok >= alphabet)
            continue;
        uint32_t run = tok - prev;
        if (run >= 3) {
            while (run > 10) {
                l1_freq[17]++;
                run = (run + 13) / 8;
            }
            l1_freq[17]++;
        } else {
            l1_freq[0] += run;
        }
        l1_freq[len]++;
        prev = tok + 1;
    }
    int ret = hydk_small_code_lengths(l1_freq, l1_len, 18, 5);
    if (ret)
        return ret;
    uint32_t space = 0;
    for (int j = 0; j < 18; j++) {
        const uint32_t len = l1_len[order[j]];
        hks_put(s, lol_bits[len], lol_len[len]);
        if (len)
            space += 32u >> len;
        if (space >= 32)
            break;
    }
    if (space && space != 32)
        return 4;
    ret = hydk_small_codes(l1_len, 18, l1_bits);
    if (ret)
        return ret;
    space = 0;
    prev = 0;
    for (uint32_t ci = 0; ci < HYDK_LF_CODES; ci++) {
        const uint32_t len = lengths[ci], tok = hks_lf_token(ci);
        if (!len || tok >= alphabet)
            continue;
        hks_zero_run(s, l1_bits, l1_len, tok - prev);
        hks_put(s, l1_bits[len], l1_len[len]);
        prev = tok + 1;
        space += 32768u >> len;
        if (space == 32768)
            return 0; /* the code is complete: nothing behind it is sent */
    }
    /* an incomplete code: the zeros up to the end of the alphabet follow (entropy.c:803) */
    hks_zero_run(s, l1_bits, l1_len, alphabet - prev);
    return 0;
}

HYDK_HD int hydk_lf_prefix_codes(HydkSink *s, const uint8_t *lengths, uint32_t alphabet0, uint32_t run_pairs) {
    const uint32_t alphabet[2] = {alphabet0, run_pairs ? 2u : 0u};
    for (int c = 0; c < 2; c++) {
        if (alphabet[c] <= 1) {
            hks_put(s, 0, 1);
            continue;
        }
        hks_put(s, 1, 1);
        const int n = hks_ilog2(alphabet[c] - 1u);
        hks_put(s, (uint32_t)n, 4);
        hks_put(s, alphabet[c] - 1u, (uint32_t)n);
    }
    for (int c = 0; c < 2; c++) {
        const uint32_t n = alphabet[c];
        if (n <= 1)
            continue;
        uint32_t used = 0, psym[4] = {0, 0, 0, 0}, plen[4] = {0, 0, 0, 0};
        if (c == 0) {
            for (uint32_t ci = 0; ci < HYDK_LF_CODES; ci++) {
                const uint32_t tok = hks_lf_token(ci);
                if (!lengths[ci] || tok >= n)
                    continue;
                if (used < 4) {
                    psym[used] = tok;
                    plen[used] = lengths[ci];
                }
                if (++used > 4)
                    break;
            }
        } /* the distance cluster only ever sees one token: no lengths, the lone-symbol form below */
        if (used > 4) {
            const int ret = hks_complex_lengths(s, lengths, n);
            if (ret)
                return ret;
            continue;
        }
        if (!used) {
            used = 1;
            psym[0] = n - 1;
        }
        hks_put(s, 1, 2); /* hskip = 1: up to four symbols */
        hks_put(s, used - 1, 2);
#define HKS_SWAP(a, b)                                                              \
    do {                                                                            \
        const uint32_t ts = psym[a], tl = plen[a];                                  \
        psym[a] = psym[b];                                                          \
        plen[a] = plen[b];                                                          \
        psym[b] = ts;                                                               \
        plen[b] = tl;                                                               \
    } while (0)
        if (used == 3 && plen[0] != 1) {
            const int o = plen[1] == 1 ? 1 : 2;
            HKS_SWAP(0, o);
        }
        int skewed = 0;
        if (used == 4) {
            for (int i = 0; i < 4; i++)
                if (plen[i] != 2)
                    skewed = 1;
            if (skewed && plen[0] != 1) {
                const int o = plen[1] == 1 ? 1 : plen[2] == 1 ? 2 : 3;
                HKS_SWAP(0, o);
            }
            if (skewed && plen[1] != 2) {
                const int o = plen[2] == 2 ? 2 : 3;
                HKS_SWAP(1, o);
            }
        }
#undef HKS_SWAP
        const int symbol_bits = hks_clog2(n);
        for (uint32_t i = 0; i < used; i++)
            hks_put(s, psym[i], (uint32_t)symbol_bits);
        if (used == 4)
            hks_put(s, (uint32_t)skewed, 1);
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------------
 * The same header written by a WAVEFRONT: hydk_lf_prefix_codes costs one lane 0.35 ms per LF group (three
 * walks over 384 code lengths, a bit at a time), and that sits on the critical path of every frame the device
 * assembles.  Here the 384 compact entries are dealt to 64 lanes, six consecutive ones each; the phases below
 * read only what earlier phases wrote, so on the device a phase is one pass of all lanes (wave_sync between
 * phases) and on the host — where the CPU tests compare it with the serial writer and with prefix.c — a loop
 * over the lanes.  Scans go through the scratch arrays: 64 short serial sums per lane, no cross-lane intrinsics.
 * ------------------------------------------------------------------------------------------- */
typedef struct HydkLfHeadScratch {
    uint32_t last_tok[64]; /* per lane: token + 1 of its last coded entry (0: it has none) */
    uint32_t bits[64];     /* per lane: bits its entries take in the complex form */
    uint32_t space[64];    /* per lane: Kraft weights (32768 >> length) of its entries */
    uint32_t nonzero[64];  /* per lane: coded entries */
    uint32_t l1_freq[18], l1_len[18], l1_bits[18];
    uint32_t used, p0, err;
    HydkSmallNode nodes[2 * HYDK_SMALL_N - 1]; /* workspace of the 18-symbol code construction */
    int32_t stack[2 * HYDK_SMALL_N];
} HydkLfHeadScratch;

#if defined(__HIP_DEVICE_COMPILE__)
#define HKS_LANES(l) for (int l = (int)threadIdx.x, hks_once_ = 1; hks_once_; hks_once_ = 0)
#define HKS_SYNC()                                             \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)
#define HKS_ATOMIC_ADD(p, v) atomicAdd((p), (v))
#else
#define HKS_LANES(l) for (int l = 0; l < 64; l++)
#define HKS_SYNC() \
    do {           \
    } while (0)
#define HKS_ATOMIC_ADD(p, v) (*(p) += (v))
#endif

/* the (value, width) one coded entry of the complex form sends: its zero run, then its length's code */
HYDK_HD uint32_t hks_entry_bits(const uint32_t *l1_bits, const uint32_t *l1_len, uint32_t run, uint32_t len, uint64_t *value) {
    HydkSink one = {(uint32_t *)0, 0, 64, 0, 0};
    uint32_t w[2] = {0, 0};
    one.w = w;
    hks_zero_run(&one, l1_bits, l1_len, run);
    hks_put(&one, l1_bits[len], l1_len[len]);
    *value = (uint64_t)w[0] | ((uint64_t)w[1] << 32);
    return (uint32_t)one.pos; /* <= 5 x (5 + 3) + 5 = 45 */
}

/* Called by all 64 lanes of one wavefront (device) / once (host).  `words` (zeroed, `cap_bits` long) already holds
 * `start` bits; lengths[] is the compact array of hydk_lf_prefix_codes.  Returns the bit position behind the header
 * through *end (valid for every lane / the caller) and 0, or a non-zero error. */
HYDK_HD int hydk_lf_prefix_codes_wave(uint32_t *words, uint64_t cap_bits, uint64_t start, const uint8_t *lengths,
                                      uint32_t alphabet0, uint32_t run_pairs, HydkLfHeadScratch *S, uint64_t *end) {
    /* phase 1: what every lane holds */
    HKS_LANES(l) {
        uint32_t last = 0, nz = 0, space = 0;
        for (uint32_t k = 0; k < 6; k++) {
            const uint32_t ci = (uint32_t)l * 6u + k, len = lengths[ci], tok = hks_lf_token(ci);
            if (!len || tok >= alphabet0)
                continue;
            last = tok + 1;
            nz++;
            space += 32768u >> len;
        }
        S->last_tok[l] = last;
        S->nonzero[l] = nz;
        S->space[l] = space;
        if (l < 18)
            S->l1_freq[l] = 0;
        if (l == 0)
            S->err = 0;
    }
    HKS_SYNC();
    uint32_t used_all = 0;
    for (int i = 0; i < 64; i++)
        used_all += S->nonzero[i];
    if (used_all <= 4 || alphabet0 <= 1) {
        /* at most four symbols: the simple form, a handful of fields — one lane writes the whole header */
        HKS_LANES(l) {
            if (l == 0) {
                HydkSink sink = {words, start, cap_bits, 0, 0};
                const int ret = hydk_lf_prefix_codes(&sink, lengths, alphabet0, run_pairs);
                S->err = ret ? (uint32_t)ret : sink.overflow ? 100u : 0u;
                S->p0 = (uint32_t)sink.pos;
            }
        }
        HKS_SYNC();
        *end = S->p0;
        return (int)S->err;
    }
    /* phase 2: the histogram of the 18-symbol alphabet (zero runs, code lengths) */
    HKS_LANES(l) {
        uint32_t prev = 0;
        for (int i = 0; i < l; i++)
            prev = S->last_tok[i] ? S->last_tok[i] : prev;
        for (uint32_t k = 0; k < 6; k++) {
            const uint32_t ci = (uint32_t)l * 6u + k, len = lengths[ci], tok = hks_lf_token(ci);
            if (!len || tok >= alphabet0)
                continue;
            uint32_t run = tok - prev;
            if (run >= 3) {
                uint32_t n17 = 1;
                while (run > 10) {
                    n17++;
                    run = (run + 13) / 8;
                }
                HKS_ATOMIC_ADD(&S->l1_freq[17], n17);
            } else if (run) {
                HKS_ATOMIC_ADD(&S->l1_freq[0], run);
            }
            HKS_ATOMIC_ADD(&S->l1_freq[len], 1u);
            prev = tok + 1;
        }
    }
    HKS_SYNC();
    /* phase 3: one lane — alphabet sizes, hskip, the code of code lengths */
    HKS_LANES(l) {
        if (l == 0) {
            const uint8_t order[18] = {1, 2, 3, 4, 0, 5, 17, 6, 16, 7, 8, 9, 10, 11, 12, 13, 14, 15};
            const uint8_t lol_bits[6] = {0, 7, 3, 2, 1, 15}, lol_len[6] = {2, 4, 3, 2, 2, 4};
            HydkSink sink = {words, start, cap_bits, 0, 0};
            const uint32_t alphabet[2] = {alphabet0, run_pairs ? 2u : 0u};
            for (int c = 0; c < 2; c++) {
                if (alphabet[c] <= 1) {
                    hks_put(&sink, 0, 1);
                    continue;
                }
                hks_put(&sink, 1, 1);
                const int n = hks_ilog2(alphabet[c] - 1u);
                hks_put(&sink, (uint32_t)n, 4);
                hks_put(&sink, alphabet[c] - 1u, (uint32_t)n);
            }
            hks_put(&sink, 0, 2); /* hskip = 0 */
            for (int i = 0; i < 18; i++)
                S->l1_bits[i] = 0;
            /* everything this lane indexes by a computed position stays in the scratch structure (LDS on the device) */
            int ret = hydk_small_code_lengths_ws(S->l1_freq, S->l1_len, 18, 5, S->nodes, S->stack);
            uint32_t space = 0;
            for (int j = 0; j < 18 && !ret; j++) {
                const uint32_t len = S->l1_len[order[j]];
                hks_put(&sink, lol_bits[len], lol_len[len]);
                if (len)
                    space += 32u >> len;
                if (space >= 32)
                    break;
            }
            if (!ret && space && space != 32)
                ret = 4;
            if (!ret)
                ret = hydk_small_codes(S->l1_len, 18, S->l1_bits);
            S->err = ret ? (uint32_t)ret : sink.overflow ? 100u : 0u;
            S->p0 = (uint32_t)sink.pos;
        }
    }
    HKS_SYNC();
    if (S->err) {
        *end = S->p0;
        return (int)S->err;
    }
    /* phase 4: how many bits each lane's entries take.  A code is complete where its Kraft sum reaches 32768:
     * nothing behind that entry is sent (entropy.c:797-799) — for the complete codes the LF coder builds that is
     * its last entry */
    HKS_LANES(l) {
        uint32_t prev = 0, before = 0;
        for (int i = 0; i < l; i++) {
            prev = S->last_tok[i] ? S->last_tok[i] : prev;
            before += S->space[i];
        }
        uint32_t bits = 0;
        for (uint32_t k = 0; k < 6 && before < 32768u; k++) {
            const uint32_t ci = (uint32_t)l * 6u + k, len = lengths[ci], tok = hks_lf_token(ci);
            if (!len || tok >= alphabet0)
                continue;
            uint64_t v;
            bits += hks_entry_bits(S->l1_bits, S->l1_len, tok - prev, len, &v);
            before += 32768u >> len;
            prev = tok + 1;
        }
        S->bits[l] = bits;
    }
    HKS_SYNC();
    /* phase 5: every lane writes its entries at its offset */
    HKS_LANES(l) {
        uint32_t prev = 0, before = 0;
        uint64_t pos = S->p0;
        for (int i = 0; i < l; i++) {
            prev = S->last_tok[i] ? S->last_tok[i] : prev;
            before += S->space[i];
            pos += S->bits[i];
        }
        HydkSink sink = {words, pos, cap_bits, 0, 1};
        for (uint32_t k = 0; k < 6 && before < 32768u; k++) {
            const uint32_t ci = (uint32_t)l * 6u + k, len = lengths[ci], tok = hks_lf_token(ci);
            if (!len || tok >= alphabet0)
                continue;
            uint64_t v;
            const uint32_t n = hks_entry_bits(S->l1_bits, S->l1_len, tok - prev, len, &v);
            hks_put64(&sink, v, n);
            before += 32768u >> len;
            prev = tok + 1;
        }
        if (sink.overflow)
            S->err = 100u;
    }
    HKS_SYNC();
    /* phase 6: one lane — the zeros behind an incomplete code, then the distance cluster's code */
    HKS_LANES(l) {
        if (l == 0) {
            uint64_t pos = S->p0;
            uint32_t prev = 0, space = 0;
            for (int i = 0; i < 64; i++) {
                pos += S->bits[i];
                prev = S->last_tok[i] ? S->last_tok[i] : prev;
                space += S->space[i];
            }
            HydkSink sink = {words, pos, cap_bits, 0, 0};
            if (space < 32768u)
                hks_zero_run(&sink, S->l1_bits, S->l1_len, alphabet0 - prev);
            if (run_pairs) { /* alphabet 2, no lengths: the lone-symbol form names symbol 1 in one bit */
                hks_put(&sink, 1, 2);
                hks_put(&sink, 0, 2);
                hks_put(&sink, 1, 1);
            }
            if (sink.overflow)
                S->err = 100u;
            S->p0 = (uint32_t)sink.pos;
        }
    }
    HKS_SYNC();
    *end = S->p0;
    return (int)S->err;
}

/* ---------------------------------------------------------------------------------------------
 * one TOC entry (encoder.c:117-120,994-1003): value and width; width 0 = the size cannot be signalled
 * ------------------------------------------------------------------------------------------- */
HYDK_HD uint32_t hydk_toc_entry(uint64_t size, uint64_t *value) {
    const uint32_t offset[4] = {0, 1024, 17408, 4211712}, bits[4] = {10, 14, 22, 30};
    for (uint32_t i = 0; i < 4; i++) {
        if (size >= offset[i] && size - offset[i] <= (((uint64_t)1 << bits[i]) - 1)) {
            *value = ((size - offset[i]) << 2) | i;
            return bits[i] + 2;
        }
    }
    *value = 0;
    return 0;
}

#endif /* HYDK_SECTIONS_H_ */
