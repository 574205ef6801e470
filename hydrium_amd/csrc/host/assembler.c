/*
 * assembler.c — host side of the device-resident frame assembler (include/hydrium_amd.h, hydamd_assembler_*).
 *
 * The kernels of csrc/hip/assemble.hip build a whole one-frame codestream from shard blobs in device memory.
 * What they cannot know from the blobs is written here, once per frame shape, with the same host functions
 * hyd_send_tile's own assembly uses (frame.c / prefix.c): the file header, the frame header with the
 * Lehmer-coded TOC permutation of the send order (reference encoder.c:241-435), LFGlobal (encoder.c:510-537),
 * the constant fields that open every LF group section (encoder.c:539-570, entropy.c:546-575) and the
 * geometry-only sub-streams that close it (encoder.c:598-626), and HFGlobal's fields in front of its
 * histograms (encoder.c:959-966, entropy.c:980-999).  The PLAN (csrc/hip/hydk_assemble.h) carries those bytes
 * and the expected order of LF groups to the device; it is rebuilt only when the description changes.
 */
#include <stdlib.h>
#include <string.h>

#include "../hip/hydk_assemble.h"
#include "bitio.h"
#include "frame.h"
#include "hydrium_amd.h"
#include "libhydrium/libhydrium.h"

int hyd_internal_file_header(const HYDImageMetadata *md, const uint8_t *icc, size_t icc_size, HydBits *out, const char **err);
const HydBits *hyd_internal_lf_tail(size_t vbw, size_t vbh);

struct HydAmdAssembler {
    HydkAsm *dev;
    const char *error;
    /* description of the current plan (to recognise a repeated call) */
    uint8_t *key;
    size_t key_len;
};

typedef struct Buf {
    uint8_t *p;
    size_t len, cap;
    int failed;
} Buf;

static size_t buf_reserve(Buf *b, size_t n) { /* returns the 16-byte aligned offset of n fresh zero bytes */
    const size_t at = (b->len + 15) & ~(size_t)15;
    const size_t need = at + ((n + 15) & ~(size_t)15) + 16;
    if (need > b->cap) {
        size_t ncap = b->cap ? b->cap : 4096;
        while (ncap < need)
            ncap *= 2;
        uint8_t *np = realloc(b->p, ncap);
        if (!np) {
            b->failed = 1;
            return 0;
        }
        memset(np + b->cap, 0, ncap - b->cap);
        b->p = np;
        b->cap = ncap;
    }
    b->len = at + n;
    return at;
}

/* a bit string (whole bytes + pending bits of a HydBits) as zero-padded words; returns its offset, *bits its length */
static size_t buf_add_bits(Buf *b, const HydBits *src, uint32_t *bits) {
    const size_t nbytes = src->len + (size_t)((src->nacc + 7) >> 3);
    const size_t at = buf_reserve(b, nbytes ? nbytes : 1);
    if (b->failed)
        return 0;
    memset(b->p + at, 0, (nbytes + 15) & ~(size_t)15);
    if (src->len)
        memcpy(b->p + at, src->data, src->len);
    uint64_t acc = src->acc;
    if (src->nacc < 64)
        acc &= (UINT64_C(1) << src->nacc) - 1;
    for (int i = 0; i * 8 < src->nacc; i++)
        b->p[at + src->len + (size_t)i] = (uint8_t)(acc >> (8 * i));
    *bits = (uint32_t)(src->len * 8 + (size_t)src->nacc);
    return at;
}

#define AFAIL(a, code, msg) ((a)->error = (msg), (code))

HYDRIUM_EXPORT HydAmdAssembler *hydamd_assembler_create(int device, int *status) {
    HydAmdAssembler *a = calloc(1, sizeof(*a));
    int st = a ? hydk_asm_create(device, &a->dev) : HYD_NOMEM;
    if (st != HYD_OK) {
        free(a);
        a = NULL;
    }
    if (status)
        *status = st;
    return a;
}

HYDRIUM_EXPORT void hydamd_assembler_destroy(HydAmdAssembler *a) {
    if (!a)
        return;
    hydk_asm_destroy(a->dev);
    free(a->key);
    free(a);
}

HYDRIUM_EXPORT const char *hydamd_assembler_error(HydAmdAssembler *a) {
    if (!a)
        return "null assembler";
    return a->error ? a->error : hydk_asm_error(a->dev);
}

HYDRIUM_EXPORT int hydamd_assembler_plan(HydAmdAssembler *a, const HYDImageMetadata *md, int write_header, int is_last,
                                         size_t nblobs, const uint32_t *blob_slots, const uint32_t *lf_ids, const uint8_t *icc,
                                         size_t icc_size) {
    if (!a || !md || !blob_slots || !lf_ids || !nblobs)
        return a ? AFAIL(a, HYD_API_ERROR, "null argument") : HYD_API_ERROR;
    a->error = NULL;
    if (nblobs > HYDK_ASM_MAX_BLOBS)
        return AFAIL(a, HYD_API_ERROR, "too many blobs for one frame");
    if (md->tile_size_shift_x >= 0 && md->tile_size_shift_y >= 0)
        return AFAIL(a, HYD_API_ERROR, "the device-side assembler builds one-frame images");
    if (!md->width || !md->height || md->width > (1u << 30) || md->height > (1u << 30))
        return AFAIL(a, HYD_API_ERROR, "width or height out of bounds");
    const size_t W = md->width, H = md->height;
    const size_t lfx = (W + 2047) >> 11, lfy = (H + 2047) >> 11, nlf = lfx * lfy;
    const size_t fgx = (W + 255) >> 8, fgy = (H + 255) >> 8, fg = fgx * fgy;
    size_t nslots = 0;
    for (size_t b = 0; b < nblobs; b++) {
        if (!blob_slots[b]) /* no LF group's workgroup would check (and report on) such a blob */
            return AFAIL(a, HYD_API_ERROR, "a blob without LF groups");
        nslots += blob_slots[b];
    }
    if (nslots != nlf || nlf > HYDAMD_MAX_LF_GROUPS || nlf == 128)
        return AFAIL(a, HYD_API_ERROR, "a frame needs every one of its LF groups (at most 255, not 128)");
    if (fg < 2)
        return AFAIL(a, HYD_API_ERROR, "a frame of one group is a single bit-contiguous section: assemble it on the host");

    /* the same description as last time? */
    const size_t key_len = sizeof(*md) + 2 * sizeof(int) + sizeof(size_t) * 2 + nblobs * sizeof(uint32_t) + nslots * sizeof(uint32_t) + icc_size;
    uint8_t *key = malloc(key_len ? key_len : 1);
    if (!key)
        return AFAIL(a, HYD_NOMEM, "out of memory");
    {
        uint8_t *k = key;
        memset(k, 0, sizeof(*md));
        HYDImageMetadata m = {0};
        m.width = md->width;
        m.height = md->height;
        m.linear_light = md->linear_light;
        m.tile_size_shift_x = md->tile_size_shift_x;
        m.tile_size_shift_y = md->tile_size_shift_y;
        memcpy(k, &m, sizeof(m));
        k += sizeof(m);
        memcpy(k, &write_header, sizeof(int));
        k += sizeof(int);
        memcpy(k, &is_last, sizeof(int));
        k += sizeof(int);
        memcpy(k, &nblobs, sizeof(size_t));
        k += sizeof(size_t);
        memcpy(k, &icc_size, sizeof(size_t));
        k += sizeof(size_t);
        memcpy(k, blob_slots, nblobs * sizeof(uint32_t));
        k += nblobs * sizeof(uint32_t);
        memcpy(k, lf_ids, nslots * sizeof(uint32_t));
        k += nslots * sizeof(uint32_t);
        if (icc_size)
            memcpy(k, icc, icc_size);
    }
    if (a->key && a->key_len == key_len && !memcmp(a->key, key, key_len)) {
        free(key);
        return HYD_OK;
    }

    int ret = HYD_OK;
    Buf buf = {0};
    HydBits bits;
    hb_init(&bits);
    HydFrameLfg *sent = calloc(nslots, sizeof(*sent));
    uint8_t *seen = calloc(nlf, 1);
    if (!sent || !seen) {
        ret = AFAIL(a, HYD_NOMEM, "out of memory");
        goto done;
    }
    buf_reserve(&buf, sizeof(HydkAsmPlan));
    if (buf.failed) {
        ret = AFAIL(a, HYD_NOMEM, "out of memory");
        goto done;
    }
    HydkAsmPlan plan;
    memset(&plan, 0, sizeof(plan));
    plan.magic = HYDK_ASM_PLAN_MAGIC;
    plan.num_slots = (uint32_t)nslots;
    plan.num_blobs = (uint32_t)nblobs;
    plan.num_presets = (uint32_t)nlf;
    plan.frame_groups = (uint32_t)fg;
    plan.toc_n = (uint32_t)(2 + nslots + fg);

    /* the LF groups in send order: blob by blob, slot by slot */
    const size_t slots_off = buf_reserve(&buf, nslots * sizeof(HydkAsmSlot));
    const size_t pslot_off = buf_reserve(&buf, nlf * sizeof(uint32_t));
    if (buf.failed) {
        ret = AFAIL(a, HYD_NOMEM, "out of memory");
        goto done;
    }
    plan.slots_off = (uint32_t)slots_off;
    plan.preset_slot_off = (uint32_t)pslot_off;
    size_t tail_vbw[HYDK_ASM_MAX_TAILS], tail_vbh[HYDK_ASM_MAX_TAILS];
    {
        size_t s = 0, group_base = 0;
        for (size_t b = 0; b < nblobs; b++) {
            plan.blob_slots[b] = blob_slots[b];
            plan.blob_first[b] = (uint32_t)s;
            for (uint32_t i = 0; i < blob_slots[b]; i++, s++) {
                const size_t id = lf_ids[s];
                if (id >= nlf || seen[id]) {
                    ret = AFAIL(a, HYD_API_ERROR, "an LF group is missing or appears twice in the frame description");
                    goto done;
                }
                seen[id] = 1;
                HydFrameLfg *l = &sent[s];
                l->raster_id = id;
                l->x = id % lfx;
                l->y = id / lfx;
                l->width = (l->x + 1) * 2048 > W ? W - l->x * 2048 : 2048;
                l->height = (l->y + 1) * 2048 > H ? H - l->y * 2048 : 2048;
                const size_t vbw = (l->width + 7) >> 3, vbh = (l->height + 7) >> 3;
                uint32_t tail = 0;
                while (tail < plan.ntails && (tail_vbw[tail] != vbw || tail_vbh[tail] != vbh))
                    tail++;
                if (tail == plan.ntails) {
                    if (tail == HYDK_ASM_MAX_TAILS) { /* a frame has at most four LF group shapes */
                        ret = AFAIL(a, HYD_INTERNAL_ERROR, "more LF group shapes than a frame can have");
                        goto done;
                    }
                    tail_vbw[tail] = vbw;
                    tail_vbh[tail] = vbh;
                    plan.ntails++;
                }
                HydkAsmSlot rec;
                memset(&rec, 0, sizeof(rec));
                rec.blob = (uint32_t)b;
                rec.index = i;
                rec.preset = (uint32_t)id;
                rec.tail = tail;
                rec.ngroups = (uint32_t)(((l->width + 255) >> 8) * ((l->height + 255) >> 8));
                rec.group_base = (uint32_t)group_base;
                group_base += rec.ngroups;
                memcpy(buf.p + slots_off + s * sizeof(rec), &rec, sizeof(rec));
                const uint32_t s32 = (uint32_t)s;
                memcpy(buf.p + pslot_off + id * sizeof(uint32_t), &s32, sizeof(s32));
            }
        }
        if (group_base != fg) {
            ret = AFAIL(a, HYD_INTERNAL_ERROR, "group count inconsistent with the frame's geometry");
            goto done;
        }
    }

    /* file header + frame header */
    if (write_header) {
        ret = hyd_internal_file_header(md, icc, icc_size, &bits, &a->error);
        if (ret)
            goto done;
    }
    {
        HydFrameShape shape;
        memset(&shape, 0, sizeof(shape));
        shape.one_frame = 1;
        shape.image_width = shape.frame_width = W;
        shape.image_height = shape.frame_height = H;
        shape.tile_count_x = shape.tile_count_y = 8;
        shape.lfg_count = nslots;
        shape.lfg = sent;
        shape.is_last = is_last;
        ret = hyd_write_frame_header(&bits, &shape, &a->error);
        if (ret) {
            if (!a->error)
                a->error = "frame header could not be written";
            goto done;
        }
        hb_align(&bits); /* hyd_write_toc_sizes starts on a byte boundary */
        uint32_t nbits = 0;
        plan.prefix_off = (uint32_t)buf_add_bits(&buf, &bits, &nbits);
        plan.prefix_bytes = nbits >> 3;
    }
    hb_reset(&bits);
    hyd_write_lf_global(&bits);
    hb_align(&bits);
    {
        uint32_t nbits = 0;
        plan.lfglobal_off = (uint32_t)buf_add_bits(&buf, &bits, &nbits);
        plan.lfglobal_bytes = nbits >> 3;
    }
    hb_reset(&bits);
    ret = hyd_write_lf_group_fixed_head(&bits, &a->error);
    if (ret)
        goto done;
    plan.lfpre_off = (uint32_t)buf_add_bits(&buf, &bits, &plan.lfpre_bits);
    hb_reset(&bits);
    {
        int per = 0;
        ret = hyd_write_hf_global_fixed(&bits, (unsigned)nlf, fg, &per, &a->error);
        if (ret)
            goto done;
        plan.clusters_per_preset = (uint32_t)per;
        plan.hfpre_off = (uint32_t)buf_add_bits(&buf, &bits, &plan.hfpre_bits);
    }
    for (uint32_t t = 0; t < plan.ntails; t++) {
        const HydBits *tail = hyd_internal_lf_tail(tail_vbw[t], tail_vbh[t]);
        if (tail) {
            plan.tail_off[t] = (uint32_t)buf_add_bits(&buf, tail, &plan.tail_bits[t]);
            continue;
        }
        /* the process-wide cache of tails is full (it holds 32 shapes): code this one here */
        hb_reset(&bits);
        ret = hyd_write_lf_group_tail(&bits, tail_vbw[t], tail_vbh[t], &a->error);
        if (ret || bits.failed) {
            ret = AFAIL(a, ret ? ret : HYD_NOMEM, "LF group tail could not be coded");
            goto done;
        }
        plan.tail_off[t] = (uint32_t)buf_add_bits(&buf, &bits, &plan.tail_bits[t]);
    }
    if (buf.failed || bits.failed) {
        ret = AFAIL(a, HYD_NOMEM, "out of memory");
        goto done;
    }
    buf.len = (buf.len + 15) & ~(size_t)15;
    plan.total_bytes = (uint32_t)buf.len;
    memcpy(buf.p, &plan, sizeof(plan));
    ret = hydk_asm_set_plan(a->dev, buf.p, buf.len);
    free(a->key); /* on failure the device's plan may be half-written: whatever comes next is planned afresh */
    a->key = NULL;
    a->key_len = 0;
    if (!ret) {
        a->key = key;
        a->key_len = key_len;
        key = NULL;
    }
done:
    free(key);
    free(sent);
    free(seen);
    free(buf.p);
    hb_free(&bits);
    return ret;
}

HYDRIUM_EXPORT int hydamd_assembler_run(HydAmdAssembler *a, const void *const *blobs_dev, const size_t *blob_caps, void *hip_stream,
                                        void *out, size_t out_cap) {
    if (!a || !blobs_dev || !blob_caps)
        return a ? AFAIL(a, HYD_API_ERROR, "null argument") : HYD_API_ERROR;
    a->error = NULL;
    if (!a->key)
        return AFAIL(a, HYD_API_ERROR, "hydamd_assembler_plan has not described the frame yet");
    uint64_t caps[HYDK_ASM_MAX_BLOBS];
    size_t nblobs = 0;
    memcpy(&nblobs, a->key + sizeof(HYDImageMetadata) + 2 * sizeof(int), sizeof(size_t));
    for (size_t b = 0; b < nblobs; b++)
        caps[b] = blob_caps[b];
    return hydk_asm_run(a->dev, blobs_dev, caps, hip_stream, out, out_cap);
}

HYDRIUM_EXPORT int hydamd_assembler_result(HydAmdAssembler *a, size_t *size) {
    if (!a || !size)
        return HYD_API_ERROR;
    uint64_t n = 0;
    uint32_t err = 0;
    hydk_asm_result(a->dev, &n, &err);
    *size = (size_t)n;
    a->error = NULL;
    if (!err)
        return n ? HYD_OK : AFAIL(a, HYD_INTERNAL_ERROR, "the assembler has not produced a frame");
    if (err & HYDK_ASM_E_NAN)
        return AFAIL(a, HYD_API_ERROR, "Invalid NaN Float");
    if (err & HYDK_ASM_E_RETRY)
        return AFAIL(a, HYD_API_ERROR, "a blob is incomplete (its frame outgrew a buffer): rerun that shard");
    if (err & HYDK_ASM_E_SPACE)
        return AFAIL(a, HYD_NEED_MORE_OUTPUT, "the frame is larger than the output buffer (size = bytes needed)");
    if (err & (HYDK_ASM_E_BLOB | HYDK_ASM_E_SLOT))
        return AFAIL(a, HYD_API_ERROR, "malformed LF-group blob");
    return AFAIL(a, HYD_INTERNAL_ERROR, "frame assembly failed on the device");
}

HYDRIUM_EXPORT int hydamd_assembler_read(HydAmdAssembler *a, uint8_t *dst, size_t capacity) {
    if (!a || !dst)
        return HYD_API_ERROR;
    a->error = NULL;
    return hydk_asm_read(a->dev, dst, capacity);
}
