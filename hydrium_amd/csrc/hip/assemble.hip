/*
 * assemble.hip — a frame put together where its sections are: on the GPU.
 *
 * What it replaces (file:line relative to /root/reference/src/libhydrium/): the closing part of
 * hyd_encode_xyb_buffer — LF group sections around their coefficient streams (encoder.c:539-629 with the
 * stream header of entropy.c:835-927), HFGlobal's histograms (encoder.c:959-967, entropy.c:303-369), the TOC
 * (encoder.c:992-1005) and the concatenation of all sections behind the frame header (encoder.c:968-1005).
 * The host used to do this from results read back piece by piece: 5.6 ms per 16384 x 16384 frame against
 * 1.9 ms of kernels.  Here the input is the shard blobs as hydamd_export_frame leaves them in device
 * memory (one per GPU that coded LF groups of the frame, gathered by RCCL or exported locally), the
 * output the finished codestream in one buffer, and the host contributes a PLAN (hydk_assemble.h): the
 * bytes that do not depend on the pixels.
 *
 *   k_asm_prepare   one launch of (LF groups + 1) workgroups:
 *       asm_slot      one per LF group: checks its blob and slot record, writes the bits in front of the LF
 *                     coefficient symbols (constant fields from the plan + alphabet sizes and prefix codes, written
 *                     by a wavefront, hydk_sections.h) and the TOC sizes of its sections
 *       asm_hfglobal  the one behind them: the ANS histograms of every cluster, each written by a thread at the
 *                     bit offset a prefix sum gives it
 *       asm_layout    whichever workgroup finishes last: TOC entries (same scheme), where every section goes, the
 *                     frame's size
 *   k_asm_copy      the frame as a list of PIECES, each the concatenation of up to three bit strings at a
 *                   byte offset; every output word is composed from its piece's strings (funnel shifts)
 *                   and stored once — a word that two pieces share is written byte by byte, so nothing is
 *                   zeroed beforehand and nothing is ORed
 *
 * Frames of a single group are one bit-contiguous section (encoder.c:837-850,968-981 guards): they stay
 * with the host assembler (hydamd_frame_from_blobs).
 */
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <new>

#include "../../../include/hydrium_amd.h"
#include "hydk_assemble.h"
#include "hydk_common.h"
#include "hydk_sections.h"

#define ST_OK 0
#define ST_NOMEM (-13)
#define ST_API_ERROR (-14)
#define ST_INTERNAL_ERROR (-15)

namespace {

constexpr int kHeadWords = 640;         /* bits in front of an LF group's symbols: <= 384 x 45 + fixed fields */
constexpr int kHfgWords = 40 * 1024;    /* HFGlobal: <= 256 histograms of <= 73 words + the cluster map */
constexpr int kTocWords = 18 * 1024;    /* <= 16579 entries of <= 32 bits */
constexpr int kMaxPieces = 8 + HYDAMD_MAX_LF_GROUPS + HYDK_ASM_MAX_BLOBS;
constexpr int kCopyBlocks = 1024;
constexpr uint32_t kBlobMagic = 0x42445948u;

struct Piece {
    uint64_t dst, nbytes;
    const uint32_t *src[3];
    uint64_t nbits[3];
};

struct BlobArgs {
    const uint8_t *p[HYDK_ASM_MAX_BLOBS];
    uint64_t cap[HYDK_ASM_MAX_BLOBS];
};

struct Scratch { /* device pointers */
    uint32_t *head;       /* [slots][kHeadWords] */
    uint32_t *head_bits;  /* [slots] */
    uint64_t *sizes;      /* [toc_n] section sizes in physical (TOC) order */
    uint64_t *slot_hf;    /* [slots] bytes of each LF group's HF sections */
    uint32_t *hfg;        /* [kHfgWords] */
    uint32_t *toc;        /* [kTocWords] */
    Piece *pieces;        /* [kMaxPieces] */
    uint32_t *npieces;    /* [1] */
    uint32_t *err;        /* [1] */
    uint32_t *done;       /* [1] workgroups of k_asm_prepare that have finished their part */
    uint64_t *result;     /* [2] size, error */
};

/* a blob whose two byte strings stay where the context keeps them (hydamd_export_frame_owned): header.lf_coded carries
 * this mark and header.reserved[1..4] the device addresses of the packed LF streams and of the packed HF sections */
constexpr uint32_t kLfCodedView = 0x101u;
__device__ __forceinline__ const uint8_t *blob_lf_bytes(const uint8_t *blob) {
    const HydAmdBlobHeader *h = (const HydAmdBlobHeader *)blob;
    if (h->lf_coded == kLfCodedView)
        return (const uint8_t *)(((uint64_t)h->reserved[2] << 32) | h->reserved[1]);
    return blob + sizeof(HydAmdBlobHeader) + (uint64_t)h->num_slots * sizeof(HydAmdBlobSlot);
}
__device__ __forceinline__ const uint8_t *blob_hf_bytes(const uint8_t *blob) {
    const HydAmdBlobHeader *h = (const HydAmdBlobHeader *)blob;
    if (h->lf_coded == kLfCodedView)
        return (const uint8_t *)(((uint64_t)h->reserved[4] << 32) | h->reserved[3]);
    return blob + (h->total_bytes - h->hf_bytes);
}
/* header sane and consistent with the plan?  (0, or HYDK_ASM_E_* bits; nothing behind the header is touched) */
__device__ __forceinline__ uint32_t blob_check(const uint8_t *blob, uint64_t cap, uint32_t want_slots) {
    const HydAmdBlobHeader *h = (const HydAmdBlobHeader *)blob;
    if (cap < sizeof(HydAmdBlobHeader) || h->magic != kBlobMagic || h->version != 1 || h->num_slots != want_slots)
        return HYDK_ASM_E_BLOB;
    uint32_t e = 0;
    if (h->status & HYDAMD_BLOB_RETRY)
        e |= HYDK_ASM_E_RETRY;
    if (h->status & 1u)
        e |= HYDK_ASM_E_NAN;
    if (e)
        return e;
    const uint64_t lf_off = sizeof(HydAmdBlobHeader) + (uint64_t)h->num_slots * sizeof(HydAmdBlobSlot);
    if (h->lf_coded == kLfCodedView) {
        if (h->total_bytes != lf_off || lf_off > cap || !(h->reserved[1] | h->reserved[2]) || !(h->reserved[3] | h->reserved[4]) ||
            ((h->reserved[1] | h->reserved[3]) & 15u))
            return HYDK_ASM_E_BLOB;
        return 0;
    }
    const uint64_t hf_off = (lf_off + h->lf_bytes + 15ull) & ~15ull;
    if (h->lf_coded != 1 || h->total_bytes > cap || lf_off > h->total_bytes || h->lf_bytes > h->total_bytes || h->hf_bytes > h->total_bytes ||
        hf_off + h->hf_bytes != h->total_bytes)
        return HYDK_ASM_E_BLOB;
    return 0;
}

__device__ __forceinline__ const HydkAsmPlan *plan_of(const uint8_t *plan) { return (const HydkAsmPlan *)plan; }

/* ---- one LF group (workgroup s of k_asm_prepare, 256 threads; the header itself is one wavefront's work) ---- */
__device__ void asm_slot(const uint8_t *__restrict__ planb, const BlobArgs &blobs, const Scratch &S, int s) {
    const HydkAsmPlan *plan = plan_of(planb);
    const int t = threadIdx.x;
    const HydkAsmSlot sl = ((const HydkAsmSlot *)(planb + plan->slots_off))[s];
    __shared__ uint32_t s_head[kHeadWords];
    __shared__ uint8_t s_len[HYDK_LF_CODES];
    __shared__ HydkLfHeadScratch s_scratch;
    __shared__ uint32_t s_bits, s_err;
    const uint8_t *blob = blobs.p[sl.blob];
    const HydAmdBlobHeader *h = (const HydAmdBlobHeader *)blob;
    uint32_t e = blob_check(blob, blobs.cap[sl.blob], plan->blob_slots[sl.blob]);
    if (!e && sl.index >= h->num_slots)
        e = HYDK_ASM_E_BLOB;
    const HydAmdBlobSlot *rec = (const HydAmdBlobSlot *)(blob + sizeof(HydAmdBlobHeader)) + sl.index;
    if (!e) {
        const uint64_t lf_end = (uint64_t)rec->lf.offset + (((uint64_t)rec->lf.bit_count + 7) >> 3);
        if (rec->preset != sl.preset || rec->table_error || rec->lf.error || lf_end > h->lf_bytes || (rec->lf.offset & 3u) ||
            rec->lf.alphabet < 1 || rec->lf.alphabet > HYDK_LF_RUN_BASE + 128u)
            e |= HYDK_ASM_E_SLOT;
    }
    if (e) {
        if (t == 0) {
            atomicOr(S.err, e);
            S.head_bits[s] = 0;
            S.slot_hf[s] = 0;
            S.sizes[1 + s] = 0;
        }
        for (uint32_t g = t; g < sl.ngroups; g += 256)
            S.sizes[2 + plan->num_slots + sl.group_base + g] = 0;
        return;
    }
    for (int i = t; i < kHeadWords; i += 256)
        s_head[i] = 0;
    for (int i = t; i < HYDK_LF_CODES; i += 256)
        s_len[i] = rec->lf.lengths[i];
    if (t == 0)
        s_err = 0;
    __syncthreads();
    {
        /* the plan's constant bits first (whole words; the bits of the last one beyond lfpre_bits are zero), then the
         * stream header's data-dependent part, written by one wavefront (hydk_sections.h) */
        const uint32_t *pre = (const uint32_t *)(planb + plan->lfpre_off);
        for (uint32_t i = t; i < (plan->lfpre_bits + 31u) >> 5; i += 256)
            s_head[i] = pre[i];
    }
    __syncthreads();
    uint64_t hf = 0;
    uint32_t bad = 0;
    if (t < 64) {
        uint64_t end = 0;
        const int ret = hydk_lf_prefix_codes_wave(s_head, (uint64_t)kHeadWords * 32u, plan->lfpre_bits, s_len, rec->lf.alphabet,
                                                  rec->lf.run_pairs, &s_scratch, &end);
        if (t == 0) {
            if (ret)
                s_err = HYDK_ASM_E_HEAD;
            s_bits = ret ? 0u : (uint32_t)end;
        }
        /* TOC sizes of this LF group's sections */
        const uint32_t b = rec->group_bits[t];
        if ((uint32_t)t < sl.ngroups) {
            const uint64_t n = ((uint64_t)b + 7u) >> 3;
            S.sizes[2 + plan->num_slots + sl.group_base + t] = n;
            hf = n;
        } else if (b) {
            bad = 1; /* a group the frame's geometry does not have */
        }
#pragma unroll
        for (int d = 32; d; d >>= 1) {
            hf += __shfl_xor(hf, d);
            bad |= (uint32_t)__shfl_xor((int)bad, d);
        }
    }
    __syncthreads();
    const uint32_t head_bits = s_bits;
    uint32_t *dst = S.head + (size_t)s * kHeadWords;
    for (uint32_t i = t; i < (head_bits + 31u) >> 5; i += 256)
        dst[i] = s_head[i];
    if (t == 0) {
        const uint64_t sec_bits = (uint64_t)head_bits + rec->lf.bit_count + plan->tail_bits[sl.tail];
        S.head_bits[s] = head_bits;
        S.sizes[1 + s] = (sec_bits + 7) >> 3;
        S.slot_hf[s] = hf;
        const uint32_t ee = s_err | (bad ? HYDK_ASM_E_SIZE : 0u);
        if (ee)
            atomicOr(S.err, ee);
    }
}

/* block-wide exclusive prefix sum over 256 threads; returns the thread's offset, *total the sum */
__device__ __forceinline__ uint64_t block_scan256(uint64_t v, uint64_t *s_wave /* [4] */, uint64_t *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t t = __shfl_up(inc, d);
        if (lane >= d)
            inc += t;
    }
    __syncthreads(); /* s_wave may still be read from an earlier call */
    if (lane == 63)
        s_wave[wave] = inc;
    __syncthreads();
    uint64_t before = 0, all = 0;
    for (int w = 0; w < 4; w++) {
        before += w < wave ? s_wave[w] : 0;
        all += s_wave[w];
    }
    *total = all;
    return before + inc - v;
}

__device__ __forceinline__ const HydAmdBlobSlot *slot_record(const uint8_t *planb, const BlobArgs &blobs, uint32_t s) {
    const HydkAsmPlan *plan = plan_of(planb);
    const HydkAsmSlot sl = ((const HydkAsmSlot *)(planb + plan->slots_off))[s];
    return (const HydAmdBlobSlot *)(blobs.p[sl.blob] + sizeof(HydAmdBlobHeader)) + sl.index;
}

/* ---- HFGlobal (the workgroup behind the LF groups' in k_asm_prepare, 256 threads) ---- */
__device__ void asm_hfglobal(const uint8_t *__restrict__ planb, const BlobArgs &blobs, const Scratch &S) {
    const HydkAsmPlan *plan = plan_of(planb);
    __shared__ uint64_t s_wave[4];
    __shared__ uint32_t s_max;
    const int t = threadIdx.x;
    {
        /* the LF groups' workgroups run beside this one: it checks the headers it is about to follow itself */
        uint32_t e = 0;
        if ((uint32_t)t < plan->num_blobs)
            e = blob_check(blobs.p[t], blobs.cap[t], plan->blob_slots[t]);
        if (e)
            atomicOr(S.err, e);
        if (__syncthreads_or((int)e))
            return;
    }
    const uint32_t per = plan->clusters_per_preset, C = plan->num_presets * per;
    if (t == 0)
        s_max = 0;
    __syncthreads();
    uint32_t mx = 0;
    for (uint32_t s = t; s < plan->num_slots; s += 256)
        mx = max(mx, slot_record(planb, blobs, s)->running_max_alphabet);
    if (mx)
        atomicMax(&s_max, mx);
    /* pass 1: how long each histogram is */
    const uint32_t *freq = nullptr;
    uint32_t alphabet = 0;
    uint64_t nb = 0;
    if ((uint32_t)t < C) {
        const uint32_t p = (uint32_t)t / per, k = (uint32_t)t % per;
        const uint32_t slot = ((const uint32_t *)(planb + plan->preset_slot_off))[p];
        const HydAmdBlobSlot *rec = slot_record(planb, blobs, slot);
        freq = rec->freq[k];
        alphabet = rec->alphabet[k] > HYDAMD_ALPHABET ? HYDAMD_ALPHABET : rec->alphabet[k];
        HydkSink count = {nullptr, 0, ~0ull, 0, 0};
        hydk_put_ans_distribution(&count, freq, alphabet);
        nb = count.pos;
    }
    uint64_t total = 0;
    const uint64_t off = block_scan256(nb, s_wave, &total);
    const uint32_t max_alpha = s_max;
    int log_alpha = max_alpha > 1 ? hks_clog2(max_alpha) : 0;
    log_alpha = log_alpha < 5 ? 5 : log_alpha;
    const uint32_t cfg_bits = (uint32_t)hks_clog2(1u + (uint32_t)log_alpha) + 3u + 2u; /* split 4, msb 1 in clog2(5), lsb 0 in clog2(4) bits */
    const uint64_t base = (uint64_t)plan->hfpre_bits + 2u + (uint64_t)C * cfg_bits;
    const uint64_t bits = base + total;
    const uint64_t words = (bits + 31) >> 5;
    if (words > (uint64_t)kHfgWords || log_alpha > 8) {
        if (t == 0)
            atomicOr(S.err, HYDK_ASM_E_SCRATCH);
        return;
    }
    for (uint64_t i = t; i < words; i += 256)
        S.hfg[i] = 0;
    __threadfence();
    __syncthreads();
    HydkSink sink = {S.hfg, 0, (uint64_t)kHfgWords * 32u, 0, 1};
    if (t == 0) {
        const uint32_t *pre = (const uint32_t *)(planb + plan->hfpre_off);
        for (uint32_t done = 0; done < plan->hfpre_bits; done += 32)
            hks_put(&sink, pre[done >> 5], plan->hfpre_bits - done < 32 ? plan->hfpre_bits - done : 32);
        hks_put(&sink, (uint32_t)(log_alpha - 5), 2);
        S.sizes[1 + plan->num_slots] = (bits + 7) >> 3;
        S.result[1] = bits; /* scratch use: the layout kernel reads HFGlobal's bit count from here */
    }
    if ((uint32_t)t < C) {
        /* hybrid-uint configuration (4, 1, 0) of cluster t (encoder.c:908, entropy.c:169-182) */
        sink.pos = (uint64_t)plan->hfpre_bits + 2u + (uint64_t)t * cfg_bits;
        hks_put(&sink, 4, cfg_bits - 5u);
        hks_put(&sink, 1, 3);
        hks_put(&sink, 0, 2);
        sink.pos = base + off;
        hydk_put_ans_distribution(&sink, freq, alphabet);
    }
}

/* ---- TOC and layout (the workgroup of k_asm_prepare that finishes last, 256 threads) ---- */
__device__ void asm_layout(const uint8_t *__restrict__ planb, const BlobArgs &blobs, const Scratch &S, uint64_t out_cap,
                           uint64_t *h_result /* pinned host [2] */) {
    const HydkAsmPlan *plan = plan_of(planb);
    __shared__ uint64_t s_wave[4];
    const int t = threadIdx.x;
    const uint32_t n = plan->toc_n, nslots = plan->num_slots;
    uint32_t err = *S.err;
    if (err) {
        if (t == 0) {
            S.result[0] = 0;
            S.result[1] = err;
            h_result[0] = 0;
            h_result[1] = err;
        }
        return;
    }
    const uint64_t hfg_bits = S.result[1];
    if (t == 0)
        S.sizes[0] = plan->lfglobal_bytes;
    __threadfence();
    __syncthreads();
    /* TOC: entry widths, where each goes, the entries */
    const uint32_t per = (n + 255u) / 256u;
    const uint32_t lo = min(n, (uint32_t)t * per), hi = min(n, lo + per);
    uint64_t mine = 0;
    uint32_t bad = 0;
    for (uint32_t i = lo; i < hi; i++) {
        uint64_t v;
        const uint32_t w = hydk_toc_entry(S.sizes[i], &v);
        bad |= w == 0;
        mine += w;
    }
    uint64_t toc_bits = 0;
    const uint64_t start = block_scan256(mine, s_wave, &toc_bits);
    const uint64_t toc_words = (toc_bits + 31) >> 5;
    if (toc_words > (uint64_t)kTocWords)
        bad |= 2;
    if (__syncthreads_or((int)bad)) {
        if (t == 0) {
            const uint32_t e = (bad & 2) ? HYDK_ASM_E_SCRATCH : HYDK_ASM_E_SIZE;
            atomicOr(S.err, e);
            S.result[0] = 0;
            S.result[1] = e;
            h_result[0] = 0;
            h_result[1] = e;
        }
        return;
    }
    for (uint64_t i = t; i < toc_words; i += 256)
        S.toc[i] = 0;
    __threadfence();
    __syncthreads();
    {
        HydkSink sink = {S.toc, start, (uint64_t)kTocWords * 32u, 0, 1};
        for (uint32_t i = lo; i < hi; i++) {
            uint64_t v;
            const uint32_t w = hydk_toc_entry(S.sizes[i], &v);
            hks_put64(&sink, v, w);
        }
    }
    const uint64_t toc_bytes = (toc_bits + 7) >> 3;
    const uint64_t body = (uint64_t)plan->prefix_bytes + toc_bytes;
    /* LF group sections: thread t owns slot t */
    uint64_t lf_mine = (uint32_t)t < nslots ? S.sizes[1 + t] : 0, lf_total = 0;
    const uint64_t lf_off = block_scan256(lf_mine, s_wave, &lf_total);
    const uint64_t lf_base = body + plan->lfglobal_bytes;
    const uint64_t hfg_dst = lf_base + lf_total, hfg_bytes = (hfg_bits + 7) >> 3;
    /* HF sections: one piece per blob, in blob order; each blob's byte count must be what its slots add up to */
    uint64_t hf_mine = 0;
    uint32_t mismatch = 0;
    if ((uint32_t)t < plan->num_blobs) {
        const HydAmdBlobHeader *h = (const HydAmdBlobHeader *)blobs.p[t];
        hf_mine = h->hf_bytes;
        uint64_t sum = 0;
        for (uint32_t s = plan->blob_first[t]; s < plan->blob_first[t] + plan->blob_slots[t]; s++)
            sum += S.slot_hf[s];
        mismatch = sum != hf_mine;
    }
    uint64_t hf_total = 0;
    const uint64_t hf_off = block_scan256(hf_mine, s_wave, &hf_total);
    const uint64_t hf_base = hfg_dst + hfg_bytes;
    const uint64_t total = hf_base + hf_total;
    Piece *P = S.pieces;
    const Piece none = {0, 0, {nullptr, nullptr, nullptr}, {0, 0, 0}};
    if (t == 0) {
        Piece p = none;
        p.dst = 0;
        p.nbytes = plan->prefix_bytes;
        p.src[0] = (const uint32_t *)(planb + plan->prefix_off);
        p.nbits[0] = (uint64_t)plan->prefix_bytes * 8u;
        P[0] = p;
        p.dst = plan->prefix_bytes;
        p.nbytes = toc_bytes;
        p.src[0] = S.toc;
        p.nbits[0] = toc_bits;
        P[1] = p;
        p.dst = body;
        p.nbytes = plan->lfglobal_bytes;
        p.src[0] = (const uint32_t *)(planb + plan->lfglobal_off);
        p.nbits[0] = (uint64_t)plan->lfglobal_bytes * 8u;
        P[2] = p;
        p.dst = hfg_dst;
        p.nbytes = hfg_bytes;
        p.src[0] = S.hfg;
        p.nbits[0] = hfg_bits;
        P[3 + nslots] = p;
        *S.npieces = 4 + nslots + plan->num_blobs;
    }
    if ((uint32_t)t < nslots) {
        const HydkAsmSlot sl = ((const HydkAsmSlot *)(planb + plan->slots_off))[t];
        const uint8_t *blob = blobs.p[sl.blob];
        const HydAmdBlobSlot *rec = (const HydAmdBlobSlot *)(blob + sizeof(HydAmdBlobHeader)) + sl.index;
        Piece p = none;
        p.dst = lf_base + lf_off;
        p.nbytes = lf_mine;
        p.src[0] = S.head + (size_t)t * kHeadWords;
        p.nbits[0] = S.head_bits[t];
        p.src[1] = (const uint32_t *)(blob_lf_bytes(blob) + rec->lf.offset);
        p.nbits[1] = rec->lf.bit_count;
        p.src[2] = (const uint32_t *)(planb + plan->tail_off[sl.tail]);
        p.nbits[2] = plan->tail_bits[sl.tail];
        P[3 + t] = p;
    }
    if ((uint32_t)t < plan->num_blobs) {
        Piece p = none;
        p.dst = hf_base + hf_off;
        p.nbytes = hf_mine;
        p.src[0] = (const uint32_t *)blob_hf_bytes(blobs.p[t]);
        p.nbits[0] = hf_mine * 8u;
        P[4 + nslots + t] = p;
    }
    const int any_mismatch = __syncthreads_or((int)mismatch);
    if (t == 0) {
        uint32_t e = any_mismatch ? HYDK_ASM_E_SIZE : 0u;
        if (!e && total > out_cap)
            e = HYDK_ASM_E_SPACE;
        if (e)
            atomicOr(S.err, e);
        S.result[0] = e == HYDK_ASM_E_SPACE ? total : e ? 0 : total;
        S.result[1] = e;
        h_result[0] = S.result[0];
        h_result[1] = e;
    }
}

/* ---- k_asm_prepare: grid = LF groups + 1, block = 256.  Workgroup s < LF groups: that LF group (asm_slot); the one
 * behind them: HFGlobal; whichever finishes last: TOC and layout — one launch where three kernels and a memset used
 * to sit in the stream (in a pipelined loop every launch of a frame costs latency in a GPU full of other frames'
 * workgroups: export + five launches took 13 % of the frame rate for 1 % of its instructions) ---- */
__global__ __launch_bounds__(256) void k_asm_prepare(const uint8_t *__restrict__ planb, BlobArgs blobs, Scratch S, uint64_t out_cap,
                                                     uint64_t *h_result) {
    __builtin_amdgcn_s_setprio(3); /* late work of a frame whose stream holds nothing else */
    __shared__ int s_last;
    const uint32_t nslots = plan_of(planb)->num_slots;
    if (blockIdx.x < nslots)
        asm_slot(planb, blobs, S, (int)blockIdx.x);
    else
        asm_hfglobal(planb, blobs, S);
    __threadfence(); /* this workgroup's results before its tick */
    __syncthreads();
    if (threadIdx.x == 0)
        s_last = atomicAdd(S.done, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!s_last)
        return;
    __threadfence(); /* everyone else's results after the last tick */
    asm_layout(planb, blobs, S, out_cap, h_result);
    __syncthreads();
    if (threadIdx.x == 0) { /* ready for the next frame */
        *S.done = 0;
        *S.err = 0;
    }
}

/* ---- k_asm_copy ---- */
__device__ __forceinline__ uint32_t word_of(const uint32_t *w, uint64_t nbits, long long i) {
    if (i < 0 || (uint64_t)i * 32u >= nbits)
        return 0;
    uint32_t v = w[i];
    const uint64_t rem = nbits - (uint64_t)i * 32u;
    if (rem < 32)
        v &= (1u << rem) - 1u;
    return v;
}
/* bits [q, q + 32) of a bit string of nbits bits; zero outside it */
__device__ __forceinline__ uint32_t bits_at(const uint32_t *w, uint64_t nbits, long long q) {
    if (!nbits || q <= -32 || q >= (long long)nbits)
        return 0;
    const long long i = q >> 5;
    const uint32_t sh = (uint32_t)(q & 31);
    const uint32_t lo = word_of(w, nbits, i);
    if (!sh)
        return lo;
    const uint32_t hi = word_of(w, nbits, i + 1);
    return (lo >> sh) | (hi << (32u - sh));
}
/* output word W as piece p sees it (zero where p has nothing) */
__device__ __forceinline__ uint32_t piece_word(const Piece &p, uint64_t W) {
    long long q = (long long)(W * 32u) - (long long)(p.dst * 8u);
    uint32_t v = bits_at(p.src[0], p.nbits[0], q);
    if (p.nbits[1]) {
        q -= (long long)p.nbits[0];
        v |= bits_at(p.src[1], p.nbits[1], q);
    } else {
        q -= (long long)p.nbits[0];
    }
    if (p.nbits[2]) {
        q -= (long long)p.nbits[1];
        v |= bits_at(p.src[2], p.nbits[2], q);
    }
    return v;
}

__global__ __launch_bounds__(256) void k_asm_copy(Scratch S, uint8_t *__restrict__ out) {
    __shared__ uint64_t s_start[kMaxPieces];
    __shared__ uint64_t s_end[kMaxPieces];
    const uint64_t total = S.result[0];
    if (!total || S.result[1])
        return;
    const uint32_t np = *S.npieces;
    for (uint32_t i = threadIdx.x; i < np; i += 256) {
        s_start[i] = S.pieces[i].dst;
        s_end[i] = S.pieces[i].dst + S.pieces[i].nbytes;
    }
    __syncthreads();
    auto find = [&](uint64_t byte) { /* the last piece that starts at or before `byte`: empty pieces sort in front of their successor */
        uint32_t lo = 0, hi = np - 1;
        while (lo < hi) {
            const uint32_t mid = (lo + hi + 1) >> 1;
            if (s_start[mid] <= byte)
                lo = mid;
            else
                hi = mid - 1;
        }
        return lo;
    };
    const uint64_t words = (total + 3) >> 2;
    uint32_t *out32 = (uint32_t *)out;
    for (uint64_t W = (uint64_t)blockIdx.x * 256u + threadIdx.x; W < words; W += (uint64_t)gridDim.x * 256u) {
        const uint64_t b0 = W * 4u;
        const uint32_t pi = find(b0);
        if (b0 + 4 <= s_end[pi]) {
            out32[W] = piece_word(S.pieces[pi], W);
            continue;
        }
        /* a word that several sections share, or the frame's last: byte by byte, each from its own piece */
        for (uint32_t j = 0; j < 4 && b0 + j < total; j++) {
            const uint32_t pj = find(b0 + j);
            out[b0 + j] = (uint8_t)(piece_word(S.pieces[pj], W) >> (8u * j));
        }
    }
}

} // namespace

extern "C" void hydamd_host_copy(void *dst, const void *src, size_t n); /* device_api.hip */

struct HydkAsm {
    int device = 0;
    char error[256] = "";
    uint8_t *plan = nullptr; /* device copy */
    size_t plan_cap = 0;
    HydkAsmPlan hplan;       /* host copy of the header */
    bool have_plan = false;
    Scratch S = {};
    uint64_t *h_result = nullptr; /* pinned */
    uint8_t *own_out = nullptr;   /* output buffer for callers that bring none (hydk_asm_run with out == NULL) */
    size_t own_cap = 0;
    uint8_t *last_out = nullptr;  /* where the last run wrote */
    hipStream_t last_stream = nullptr; /* ... and the stream it ran in (the caller's: it must outlive hydk_asm_read of that run) */
    hipEvent_t done = nullptr;    /* the assembler's own: recorded behind every run; what a plan change and a run in another stream wait for */
    uint64_t last_cap = 0;        /* ... and how large that buffer is */
    bool ran = false;             /* `done` was recorded behind a complete run */
    bool unsettled = false;       /* a run's launches began and its event was never recorded (a launch failed in between): `done` says nothing about them */
    uint8_t *bounce = nullptr;    /* pinned: hydk_asm_read lands frames here (DMA engines), then copies to the caller's memory */
    size_t bounce_cap = 0;
};

namespace {
int afail(HydkAsm *a, int code, const char *what, hipError_t e = hipSuccess) {
    if (a) {
        if (e != hipSuccess)
            snprintf(a->error, sizeof(a->error), "%s: %s", what, hipGetErrorString(e));
        else
            snprintf(a->error, sizeof(a->error), "%s", what);
    }
    return code;
}
#define ASM_TRY(a, call)                                                                              \
    do {                                                                                              \
        hipError_t e__ = (call);                                                                      \
        if (e__ != hipSuccess)                                                                        \
            return afail(a, e__ == hipErrorOutOfMemory ? ST_NOMEM : ST_INTERNAL_ERROR, #call, e__);   \
    } while (0)
} // namespace

extern "C" {

const char *hydk_asm_error(HydkAsm *a) { return a ? a->error : "null assembler"; }

void hydk_asm_destroy(HydkAsm *a) {
    if (!a)
        return;
    (void)hipSetDevice(a->device);
    void *dev[] = {a->own_out, a->plan, a->S.head, a->S.head_bits, a->S.sizes, a->S.slot_hf, a->S.hfg, a->S.toc, a->S.pieces, a->S.npieces,
                   a->S.err, a->S.done, a->S.result};
    for (void *p : dev)
        if (p)
            (void)hipFree(p);
    if (a->h_result)
        (void)hipHostFree(a->h_result);
    if (a->bounce)
        (void)hipHostFree(a->bounce);
    if (a->done)
        (void)hipEventDestroy(a->done);
    delete a;
}

static int asm_alloc(HydkAsm *a) {
    const size_t slots = HYDAMD_MAX_LF_GROUPS;
    const size_t toc_max = 2 + slots + slots * HYDK_GROUPS_PER_LFG;
    ASM_TRY(a, hipSetDevice(a->device));
    ASM_TRY(a, hipMalloc(&a->S.head, slots * kHeadWords * sizeof(uint32_t)));
    ASM_TRY(a, hipMalloc(&a->S.head_bits, slots * sizeof(uint32_t)));
    ASM_TRY(a, hipMalloc(&a->S.sizes, toc_max * sizeof(uint64_t)));
    ASM_TRY(a, hipMalloc(&a->S.slot_hf, slots * sizeof(uint64_t)));
    ASM_TRY(a, hipMalloc(&a->S.hfg, (size_t)kHfgWords * sizeof(uint32_t)));
    ASM_TRY(a, hipMalloc(&a->S.toc, (size_t)kTocWords * sizeof(uint32_t)));
    ASM_TRY(a, hipMalloc(&a->S.pieces, (size_t)kMaxPieces * sizeof(Piece)));
    ASM_TRY(a, hipMalloc(&a->S.npieces, sizeof(uint32_t)));
    ASM_TRY(a, hipMalloc(&a->S.err, sizeof(uint32_t)));
    ASM_TRY(a, hipMalloc(&a->S.done, sizeof(uint32_t)));
    ASM_TRY(a, hipMemset(a->S.err, 0, sizeof(uint32_t)));
    ASM_TRY(a, hipMemset(a->S.done, 0, sizeof(uint32_t)));
    /* hipMemset of device memory returns before it has run, in the NULL stream, which non-blocking streams do not wait
     * for: without this wait the first frame's kernels can pass the memsets (seen as a "malformed blob" once the
     * caller's stream and the null stream sat on different hardware queues) */
    ASM_TRY(a, hipStreamSynchronize(nullptr));
    ASM_TRY(a, hipMalloc(&a->S.result, 2 * sizeof(uint64_t)));
    ASM_TRY(a, hipHostMalloc((void **)&a->h_result, 2 * sizeof(uint64_t), hipHostMallocDefault));
    a->h_result[0] = a->h_result[1] = 0;
    ASM_TRY(a, hipEventCreateWithFlags(&a->done, hipEventDisableTiming));
    return ST_OK;
}

int hydk_asm_create(int device, HydkAsm **out) {
    int n = 0;
    if (!out)
        return ST_API_ERROR;
    *out = nullptr;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n)
        return ST_INTERNAL_ERROR;
    HydkAsm *a = new (std::nothrow) HydkAsm();
    if (!a)
        return ST_NOMEM;
    a->device = device;
    const int st = asm_alloc(a);
    if (st != ST_OK) {
        hydk_asm_destroy(a);
        return st;
    }
    *out = a;
    return ST_OK;
}

int hydk_asm_set_plan(HydkAsm *a, const void *plan, size_t bytes) {
    if (!a || !plan || bytes < sizeof(HydkAsmPlan))
        return afail(a, ST_API_ERROR, "bad plan");
    const HydkAsmPlan *hp = (const HydkAsmPlan *)plan;
    if (hp->magic != HYDK_ASM_PLAN_MAGIC || hp->total_bytes != bytes || hp->num_slots < 1 || hp->num_slots > HYDAMD_MAX_LF_GROUPS ||
        hp->num_blobs < 1 || hp->num_blobs > HYDK_ASM_MAX_BLOBS || hp->num_presets * hp->clusters_per_preset > 256 ||
        hp->toc_n != 2 + hp->num_slots + hp->frame_groups || hp->ntails > HYDK_ASM_MAX_TAILS)
        return afail(a, ST_API_ERROR, "inconsistent plan");
    ASM_TRY(a, hipSetDevice(a->device));
    /* an earlier frame may still be reading the old plan: wait for IT only (a device-wide wait would stall every other
     * encoder thread's frames in a batch of differently shaped images) — by the assembler's own event, not by the caller's
     * stream handle, which may be gone by now; runs in different streams are chained behind each other (hydk_asm_run), so
     * the last run's event covers them all */
    if (a->unsettled) { /* kernels of a run that failed half-way may still be reading the old plan and scratch arrays */
        ASM_TRY(a, hipDeviceSynchronize());
        a->unsettled = false;
    }
    if (a->ran)
        ASM_TRY(a, hipEventSynchronize(a->done));
    if (bytes > a->plan_cap) {
        if (a->plan)
            (void)hipFree(a->plan);
        a->plan = nullptr;
        a->plan_cap = 0;
        ASM_TRY(a, hipMalloc(&a->plan, bytes + 16)); /* + 16: the copy kernel reads whole words */
        a->plan_cap = bytes;
    }
    ASM_TRY(a, hipMemcpy(a->plan, plan, bytes, hipMemcpyHostToDevice));
    a->hplan = *hp;
    a->have_plan = true;
    return ST_OK;
}

int hydk_asm_run(HydkAsm *a, const void *const *blobs, const uint64_t *blob_caps, void *stream, void *out, uint64_t out_cap) {
    if (!a || !a->have_plan || !blobs || !blob_caps)
        return afail(a, ST_API_ERROR, "assembler not ready");
    ASM_TRY(a, hipSetDevice(a->device));
    hipStream_t st = (hipStream_t)stream;
    if (!out) { /* the assembler's own device buffer: `out_cap` bytes if the caller names a size, else what the blobs could hold
                 * (no frame is larger than its self-contained blobs plus its headers; views need a size from the caller) */
        size_t need = (size_t)out_cap;
        if (!need) {
            need = (size_t)1 << 20;
            for (uint32_t b = 0; b < a->hplan.num_blobs; b++)
                need += (size_t)blob_caps[b];
        }
        if (need > a->own_cap) {
            ASM_TRY(a, hipStreamSynchronize(st));
            if (a->own_out)
                (void)hipFree(a->own_out);
            a->own_out = nullptr;
            a->own_cap = 0;
            ASM_TRY(a, hipMalloc(&a->own_out, need));
            a->own_cap = need;
        }
        out = a->own_out;
        out_cap = a->own_cap;
    }
    /* k_asm_copy moves whole 32-bit words of the output and 16-byte-aligned records of the blobs */
    if ((uintptr_t)out & 3u)
        return afail(a, ST_API_ERROR, "output buffer must be 4-byte aligned");
    if (a->unsettled) { /* the last run failed half-way: nothing says when its kernels are done with the scratch arrays */
        ASM_TRY(a, hipDeviceSynchronize());
        a->unsettled = false;
    }
    if (a->ran && st != a->last_stream) /* the runs share the assembler's scratch arrays: one behind the other */
        ASM_TRY(a, hipStreamWaitEvent(st, a->done, 0));
    a->last_out = (uint8_t *)out;
    a->last_cap = out_cap;
    a->last_stream = st;
    a->unsettled = true; /* until this run's event is recorded (`ran` used to be set here, before the launches) */
    BlobArgs args;
    memset(&args, 0, sizeof(args));
    for (uint32_t b = 0; b < a->hplan.num_blobs; b++) {
        if (!blobs[b])
            return afail(a, ST_API_ERROR, "null blob");
        if ((uintptr_t)blobs[b] & 15u)
            return afail(a, ST_API_ERROR, "blobs must be 16-byte aligned");
        args.p[b] = (const uint8_t *)blobs[b];
        args.cap[b] = blob_caps[b];
    }
    hipLaunchKernelGGL(k_asm_prepare, dim3(a->hplan.num_slots + 1), dim3(256), 0, st, (const uint8_t *)a->plan, args, a->S, out_cap,
                       a->h_result);
    hipLaunchKernelGGL(k_asm_copy, dim3(kCopyBlocks), dim3(256), 0, st, a->S, (uint8_t *)out);
    ASM_TRY(a, hipGetLastError());
    ASM_TRY(a, hipEventRecord(a->done, st));
    a->ran = true;
    a->unsettled = false;
    return ST_OK;
}

int hydk_asm_result(HydkAsm *a, uint64_t *size, uint32_t *err) {
    if (!a)
        return ST_API_ERROR;
    if (size)
        *size = a->h_result[0];
    if (err)
        *err = (uint32_t)a->h_result[1];
    return ST_OK;
}

/* after the stream has been synchronised: the frame's bytes, copied out of device memory.  `dst` is ordinary (pageable)
 * memory as a rule: a plain hipMemcpy into it is staged by the runtime through shader copies in 3 MB pieces (an 8K frame's
 * 12 MB: 0.6 ms, and a length that is not a multiple of four takes the same path whatever the destination).  So: one DMA
 * of the length rounded up to 256 bytes into a pinned buffer of the assembler's (56 GB/s), then the staging threads' memcpy */
int hydk_asm_read(HydkAsm *a, uint8_t *dst, size_t capacity) {
    if (!a || !dst || !a->last_out)
        return afail(a, ST_API_ERROR, "nothing to read");
    if (a->h_result[1] || !a->h_result[0] || a->h_result[0] > capacity)
        return afail(a, ST_API_ERROR, "no finished frame of that size");
    ASM_TRY(a, hipSetDevice(a->device));
    const size_t n = (size_t)a->h_result[0];
    const size_t padded = (n + 255) & ~(size_t)255;
    if (n < ((size_t)1 << 20) || padded > a->last_cap) { /* small, or no room to round up: the runtime's own path */
        ASM_TRY(a, hipMemcpy(dst, a->last_out, n, hipMemcpyDeviceToHost));
        return ST_OK;
    }
    if (padded > a->bounce_cap) {
        if (a->bounce)
            (void)hipHostFree(a->bounce);
        a->bounce = nullptr;
        a->bounce_cap = 0;
        const size_t want = padded + (padded >> 2);
        ASM_TRY(a, hipHostMalloc((void **)&a->bounce, want, hipHostMallocDefault));
        a->bounce_cap = want;
    }
    ASM_TRY(a, hipMemcpyAsync(a->bounce, a->last_out, padded, hipMemcpyDeviceToHost, a->last_stream));
    ASM_TRY(a, hipStreamSynchronize(a->last_stream));
    hydamd_host_copy(dst, a->bounce, n);
    return ST_OK;
}

int hydk_asm_debug(HydkAsm *a, uint32_t slot, uint32_t *head_bits, uint32_t *head_words, size_t head_cap, uint32_t *hfg_bits,
                   uint32_t *hfg_words, size_t hfg_cap) {
    if (!a || slot >= HYDAMD_MAX_LF_GROUPS)
        return ST_API_ERROR;
    ASM_TRY(a, hipSetDevice(a->device));
    ASM_TRY(a, hipDeviceSynchronize());
    if (head_bits)
        ASM_TRY(a, hipMemcpy(head_bits, a->S.head_bits + slot, sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (head_words)
        ASM_TRY(a, hipMemcpy(head_words, a->S.head + (size_t)slot * kHeadWords,
                             (head_cap < (size_t)kHeadWords ? head_cap : (size_t)kHeadWords) * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (hfg_bits) {
        uint64_t sz = 0;
        ASM_TRY(a, hipMemcpy(&sz, a->S.sizes + 1 + a->hplan.num_slots, sizeof(uint64_t), hipMemcpyDeviceToHost));
        *hfg_bits = (uint32_t)(sz * 8u);
    }
    if (hfg_words)
        ASM_TRY(a, hipMemcpy(hfg_words, a->S.hfg, (hfg_cap < (size_t)kHfgWords ? hfg_cap : (size_t)kHfgWords) * sizeof(uint32_t),
                             hipMemcpyDeviceToHost));
    return ST_OK;
}

} /* extern "C" */
