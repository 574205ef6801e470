/*
 * multi.c — ONE frame whose pixels already sit in HBM, on N devices of one process, from C.
 *
 * The reference codes a frame's LF groups one after another on one core: hyd_send_tile per tile
 * (libhydrium.c:172-203), per-LF-group tables with the running alphabet maximum (entropy.c:459-460), the frame
 * closed by encoder.c:928-957.  hyd_send_tile's own multi-device form (encoder.c finish_frame_multi) does that on
 * several GPUs but takes its pixels from HOST memory — 805 MB through one caller thread's staging for a 16384^2
 * frame: upload-bound at any N.  This is the same composition without the uploads: the caller's pixels are device
 * pointers, one per shard, every step between them device-side —
 *     deal LF groups in raster runs  ->  transform stage per shard  ->  alphabet floor by peer read
 *     (hydamd_alphabet_floor_from_peers)  ->  closing stage per shard  ->  every shard's blob as a view
 *     (hydamd_export_frame_owned)  ->  the assembling shard's stream waits for the others (hydamd_wait_for) and its
 *     assembler reads all blobs in place, the other devices' over xGMI, and writes the finished FILE into its HBM
 * — no RCCL, no process group, no host copy before the finished file.  Asynchronous: hydamd_encode_image_multi
 * returns with everything enqueued; hydamd_multi_result waits, and reruns what a shard that outgrew its buffers
 * invalidated.  The assembling shard is the caller's choice per frame (rotate it: the file's D2H copy then leaves
 * through a different GPU's link every time).  Peer reads verify themselves at first use of a device pair, as in
 * encoder.c; a mismatch fails the frame with the pair named (there is no host copy of the pixels to fall back on:
 * the caller owns that decision).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#include "hydrium_amd.h"

#ifndef HYDRIUM_EXPORT
#define HYDRIUM_EXPORT __attribute__((visibility("default")))
#endif

#define MULTI_MAX HYDAMD_MAX_PEERS

struct HydAmdMulti {
    int n;
    HydAmdContext *ctx[MULTI_MAX];
    int device[MULTI_MAX];
    size_t first[MULTI_MAX], slots[MULTI_MAX];
    HYDImageMetadata md;
    size_t lfx, lfy, total;
    uint32_t lf_ids[HYDAMD_MAX_LF_GROUPS], blob_slots[MULTI_MAX];
    int assembling; /* shard whose device assembles the frame in flight */
    int in_flight, have_result;
    size_t out_cap, size;
    unsigned reruns[MULTI_MAX];
    int check_view[MULTI_MAX], check_floor[MULTI_MAX], checking;
    char err[200];
};

/* peer reads seen to return what their owner wrote: [reading device][owning device] (device ids below 16; others are
 * verified every time) */
static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;
static unsigned char g_pair_ok[16][16];
static int pair_known(int reader, int owner) {
    if (reader < 0 || owner < 0 || reader >= 16 || owner >= 16)
        return 0;
    pthread_mutex_lock(&g_lock);
    const int ok = g_pair_ok[reader][owner];
    pthread_mutex_unlock(&g_lock);
    return ok;
}
static void pair_latch(int reader, int owner) {
    if (reader < 0 || owner < 0 || reader >= 16 || owner >= 16)
        return;
    pthread_mutex_lock(&g_lock);
    g_pair_ok[reader][owner] = 1;
    pthread_mutex_unlock(&g_lock);
}
static int verify_mode(void) { /* as encoder.c: HYDAMD_VERIFY_PEERS unset = first use, 1 = every frame, 0 = never */
    static int mode = -1;
    if (mode < 0) {
        const char *v = getenv("HYDAMD_VERIFY_PEERS");
        mode = !v || !*v ? 2 : *v == '0' ? 0 : 1;
    }
    return mode;
}

static int fail(HydAmdMulti *m, int code, const char *what, HydAmdContext *c) {
    const char *d = c ? hydamd_error(c) : NULL;
    snprintf(m->err, sizeof(m->err), "%s%s%s", what, d && *d ? ": " : "", d && *d ? d : "");
    return code;
}

HYDRIUM_EXPORT const char *hydamd_multi_error(HydAmdMulti *m) { return m ? m->err : "null multi-device frame"; }

HYDRIUM_EXPORT void hydamd_multi_destroy(HydAmdMulti *m) {
    if (!m)
        return;
    for (int d = 0; d < m->n; d++)
        if (m->ctx[d]) {
            (void)hydamd_sync(m->ctx[d]);
            hydamd_destroy(m->ctx[d]);
        }
    free(m);
}

HYDRIUM_EXPORT HydAmdMulti *hydamd_multi_create(int n, const int *devices, const HYDImageMetadata *md, int *status) {
    int st = HYD_API_ERROR;
    HydAmdMulti *m = NULL;
    if (n < 1 || n > MULTI_MAX || !devices || !md || !md->width || !md->height)
        goto out;
    const size_t lfx = (md->width + 2047) >> 11, lfy = (md->height + 2047) >> 11, total = lfx * lfy;
    if (total > HYDAMD_MAX_LF_GROUPS || total == 128 || total < (size_t)n) /* (128: the reference never returns, entropy.c:99) */
        goto out;
    if (n > 1 && !hydamd_peers_reachable(devices, n)) {
        st = HYD_INTERNAL_ERROR;
        goto out;
    }
    m = calloc(1, sizeof(*m));
    if (!m) {
        st = HYD_NOMEM;
        goto out;
    }
    m->n = n;
    m->md = *md;
    m->lfx = lfx;
    m->lfy = lfy;
    m->total = total;
    for (size_t g = 0; g < total; g++)
        m->lf_ids[g] = (uint32_t)g; /* raster order = send order = slot order across the shards */
    for (int d = 0; d < n; d++) {
        m->device[d] = devices[d];
        m->first[d] = (size_t)d * total / (size_t)n;
        m->slots[d] = (size_t)(d + 1) * total / (size_t)n - m->first[d];
        m->blob_slots[d] = (uint32_t)m->slots[d];
        m->ctx[d] = hydamd_create(devices[d], (int)m->slots[d], md->linear_light != 0, 0, &st);
        if (!m->ctx[d]) {
            hydamd_multi_destroy(m);
            m = NULL;
            goto out;
        }
        if ((st = hydamd_set_lf_coder(m->ctx[d], 2)) != 0 || (st = hydamd_set_rans_waves(m->ctx[d], 5)) != 0) {
            hydamd_multi_destroy(m);
            m = NULL;
            goto out;
        }
    }
    st = HYD_OK;
out:
    if (status)
        *status = st;
    return m;
}

HYDRIUM_EXPORT HydAmdContext *hydamd_multi_context(HydAmdMulti *m, int shard) {
    return m && shard >= 0 && shard < m->n ? m->ctx[shard] : NULL;
}

HYDRIUM_EXPORT int hydamd_multi_shard_lf_groups(HydAmdMulti *m, int shard, size_t *first, size_t *count) {
    if (!m || shard < 0 || shard >= m->n)
        return HYD_API_ERROR;
    if (first)
        *first = m->first[shard];
    if (count)
        *count = m->slots[shard];
    return HYD_OK;
}

/* every shard's view, the assembling shard waiting for the others, the assembly itself */
static int enqueue_assembly(HydAmdMulti *m) {
    const int a = m->assembling;
    const void *blob[MULTI_MAX];
    size_t cap[MULTI_MAX];
    int st;
    for (int d = 0; d < m->n; d++)
        if ((st = hydamd_export_frame_owned(m->ctx[d], (int)m->slots[d], &blob[d], &cap[d])) != 0)
            return fail(m, st, "export", m->ctx[d]);
    for (int d = 0; d < m->n; d++)
        if (d != a && (st = hydamd_wait_for(m->ctx[a], m->ctx[d])) != 0)
            return fail(m, st, "cross-device wait", m->ctx[a]);
    for (int d = 0; d < m->n; d++)
        if (d != a && m->check_view[d]) { /* summed where it was written and where it is about to be read */
            if ((st = hydamd_verify_enqueue(m->ctx[d], m->ctx[d], (int)m->slots[d], 0)) != 0)
                return fail(m, st, "checksum on the owning device", m->ctx[d]);
            if ((st = hydamd_verify_enqueue(m->ctx[a], m->ctx[d], (int)m->slots[d], d)) != 0)
                return fail(m, st, "checksum through peer reads", m->ctx[a]);
        }
    HydAmdAssembler *as = hydamd_context_assembler(m->ctx[a]);
    if (!as)
        return fail(m, HYD_INTERNAL_ERROR, "frame assembler could not be created", m->ctx[a]);
    if ((st = hydamd_assembler_plan(as, &m->md, 1, 1, (size_t)m->n, m->blob_slots, m->lf_ids, NULL, 0)) != 0)
        return fail(m, st, hydamd_assembler_error(as), NULL);
    if ((st = hydamd_assembler_run(as, blob, cap, hydamd_get_stream(m->ctx[a]), NULL, m->out_cap)) != 0)
        return fail(m, st, hydamd_assembler_error(as), NULL);
    return HYD_OK;
}

/* src: 3 pointers per shard, [shard][channel], in shard d's device memory: where pixel (0, 0) OF THE IMAGE would sit for
 * that shard's buffer (only the pixels of the shard's own LF groups — hydamd_multi_shard_lf_groups, raster order — are
 * read; a rank-style slab of rows [y0, y1) passes slab - y0 * row_stride).  Strides in samples, as hyd_send_tile's. */
HYDRIUM_EXPORT int hydamd_encode_image_multi(HydAmdMulti *m, const void *const *src, ptrdiff_t row_stride, ptrdiff_t pixel_stride,
                                             int sample_fmt, int assembling_shard) {
    if (!m)
        return HYD_API_ERROR;
    if (!src || assembling_shard < 0 || assembling_shard >= m->n)
        return fail(m, HYD_API_ERROR, "bad arguments", NULL);
    if (sample_fmt != HYD_UINT8 && sample_fmt != HYD_UINT16 && sample_fmt != HYD_FLOAT32)
        return fail(m, HYD_API_ERROR, "Invalid Sample Format", NULL);
    if (m->in_flight)
        return fail(m, HYD_API_ERROR, "a frame is in flight: hydamd_multi_result first", NULL);
    const ptrdiff_t ss = sample_fmt == HYD_UINT8 ? 1 : sample_fmt == HYD_UINT16 ? 2 : 4;
    const size_t W = m->md.width, H = m->md.height;
    int st;
    m->assembling = assembling_shard;
    m->have_result = 0;
    m->err[0] = 0;
    const int mode = verify_mode();
    m->checking = 0;
    for (int d = 0; d < m->n; d++) {
        m->check_view[d] = d != assembling_shard && mode && (mode == 1 || !pair_known(m->device[assembling_shard], m->device[d]));
        m->check_floor[d] = 0;
        for (int p = 0; p < d; p++)
            m->check_floor[d] |= mode && (mode == 1 || !pair_known(m->device[d], m->device[p]));
        m->checking |= m->check_view[d] | m->check_floor[d];
    }
    for (int d = 0; d < m->n; d++) {
        HydAmdContext *c = m->ctx[d];
        if (!src[3 * d] || !src[3 * d + 1] || !src[3 * d + 2])
            return fail(m, HYD_API_ERROR, "null pixel pointer", NULL);
        if ((st = hydamd_begin_frame(c, (unsigned)m->total)) != 0)
            return fail(m, st, "begin frame", c);
        for (size_t i = 0; i < m->slots[d]; i++) {
            const size_t g = m->first[d] + i, tx = g % m->lfx, ty = g / m->lfx;
            const ptrdiff_t off = ((ptrdiff_t)(ty * 2048) * row_stride + (ptrdiff_t)(tx * 2048) * pixel_stride) * ss;
            const void *p[3] = {(const char *)src[3 * d] + off, (const char *)src[3 * d + 1] + off, (const char *)src[3 * d + 2] + off};
            const size_t w = W - tx * 2048 < 2048 ? W - tx * 2048 : 2048, h = H - ty * 2048 < 2048 ? H - ty * 2048 : 2048;
            if ((st = hydamd_encode_lf_group(c, (int)i, p, row_stride, pixel_stride, sample_fmt, w, h, (unsigned)g)) != 0)
                return fail(m, st, "LF group", c);
        }
        if ((st = hydamd_run_transform(c, (int)m->slots[d])) != 0)
            return fail(m, st, "transform stage", c);
    }
    for (int d = 1; d < m->n; d++) /* shard d's tables start from the maximum over shards 0 .. d-1 (entropy.c:459-460) */
        if ((st = hydamd_alphabet_floor_from_peers(m->ctx[d], d, m->ctx)) != 0)
            return fail(m, st, "alphabet floor from the earlier shards", m->ctx[d]);
    if (!m->out_cap) {
        m->out_cap = 4096 * m->total + (256u << 10);
        for (int d = 0; d < m->n; d++)
            m->out_cap += hydamd_blob_bound(m->ctx[d], (int)m->slots[d]);
    }
    for (int d = 0; d < m->n; d++) {
        if ((st = hydamd_finish_frame(m->ctx[d], (int)m->slots[d])) != 0)
            return fail(m, st, "closing stage", m->ctx[d]);
        m->reruns[d] = hydamd_overflow_reruns(m->ctx[d]);
    }
    if ((st = enqueue_assembly(m)) != 0)
        return st;
    m->in_flight = 1;
    return HYD_OK;
}

/* waits for the frame; *size = bytes of the finished file in the assembling device's memory */
HYDRIUM_EXPORT int hydamd_multi_result(HydAmdMulti *m, size_t *size) {
    if (!m)
        return HYD_API_ERROR;
    if (m->have_result) {
        if (size)
            *size = m->size;
        return HYD_OK;
    }
    if (!m->in_flight)
        return fail(m, HYD_API_ERROR, "no frame in flight", NULL);
    const int a = m->assembling;
    int st;
    for (int attempt = 0; attempt < 4 + 2 * m->n; attempt++) {
        for (int k = 1; k <= m->n; k++) { /* a shard whose frame outgrew its buffers reruns it in here: its blob is then stale */
            const int dd = (a + k) % m->n; /* the assembling shard last: its stream carries the assembly */
            if ((st = hydamd_sync(m->ctx[dd])) != 0) {
                m->in_flight = 0;
                return fail(m, st, "shard", m->ctx[dd]);
            }
        }
        /* a shard that reran had left incomplete alphabet maxima the first time: the later shards read their floor again
         * and run again (as encoder.c finish_frame_multi) */
        int stale_from = 0;
        for (int d = 0; d < m->n; d++) {
            const unsigned now = hydamd_overflow_reruns(m->ctx[d]);
            if (now != m->reruns[d] && !stale_from && d + 1 < m->n)
                stale_from = d + 1;
            m->reruns[d] = now;
        }
        if (stale_from) {
            for (int d = stale_from; d < m->n; d++)
                if ((st = hydamd_alphabet_floor_from_peers(m->ctx[d], d, m->ctx)) != 0 || (st = hydamd_replay_frame(m->ctx[d])) != 0) {
                    m->in_flight = 0;
                    return fail(m, st, "replay behind a rerun shard", m->ctx[d]);
                }
            if ((st = enqueue_assembly(m)) != 0) {
                m->in_flight = 0;
                return st;
            }
            continue;
        }
        HydAmdAssembler *as = hydamd_context_assembler(m->ctx[a]);
        size_t sz = 0;
        st = hydamd_assembler_result(as, &sz);
        int asm_failed = 0;
        if (st) {
            const char *e = hydamd_assembler_error(as);
            if ((e && strstr(e, "incomplete")) || (st == HYD_NEED_MORE_OUTPUT && sz > m->out_cap)) {
                if (st == HYD_NEED_MORE_OUTPUT)
                    m->out_cap = sz;
                if ((st = enqueue_assembly(m)) != 0) {
                    m->in_flight = 0;
                    return st;
                }
                continue;
            }
            if (e && strstr(e, "NaN")) { /* the caller's input, whatever the peer reads did */
                m->in_flight = 0;
                return fail(m, HYD_API_ERROR, "Invalid NaN Float", NULL);
            }
            if (!m->checking) {
                m->in_flight = 0;
                return fail(m, st < HYD_ERROR_START ? st : HYD_INTERNAL_ERROR, e ? e : "GPU frame assembly failed", NULL);
            }
            asm_failed = 1; /* an assembler that read garbage through a bad peer mapping: the checks name the pair */
            snprintf(m->err, sizeof(m->err), "%s", e ? e : "GPU frame assembly failed");
        }
        for (int d = 1; d < m->n; d++)
            if (m->check_floor[d]) {
                int ok = 0;
                if ((st = hydamd_verify_floor(m->ctx[d], d, m->ctx, &ok)) != 0) {
                    m->in_flight = 0;
                    return fail(m, st, "floor verification", m->ctx[d]);
                }
                if (!ok) {
                    m->in_flight = 0;
                    snprintf(m->err, sizeof(m->err), "peer read mismatch: device %d did not see the alphabet maxima devices before it wrote (shard %d)",
                             m->device[d], d);
                    return HYD_INTERNAL_ERROR;
                }
            }
        for (int d = 0; d < m->n; d++)
            if (m->check_view[d]) {
                unsigned long long written = 0, seen = 0;
                if ((st = hydamd_verify_read(m->ctx[d], 0, &written)) != 0 || (st = hydamd_verify_read(m->ctx[a], d, &seen)) != 0) {
                    m->in_flight = 0;
                    return fail(m, st, "view verification", m->ctx[d]);
                }
                if (written != seen) {
                    m->in_flight = 0;
                    snprintf(m->err, sizeof(m->err), "peer read mismatch: device %d did not see what device %d wrote (shard %d)", m->device[a],
                             m->device[d], d);
                    return HYD_INTERNAL_ERROR;
                }
            }
        m->in_flight = 0;
        if (asm_failed)
            return HYD_INTERNAL_ERROR; /* m->err holds the assembler's message */
        for (int d = 0; d < m->n; d++) { /* these peer reads returned what their owners wrote: trusted from here on */
            if (m->check_view[d])
                pair_latch(m->device[a], m->device[d]);
            for (int p = 0; p < d && m->check_floor[d]; p++)
                pair_latch(m->device[d], m->device[p]);
        }
        m->size = sz;
        m->have_result = 1;
        if (size)
            *size = sz;
        return HYD_OK;
    }
    m->in_flight = 0;
    return fail(m, HYD_INTERNAL_ERROR, "frame still does not fit after enlarging its buffers", NULL);
}

/* the finished file to host memory (one copy from the assembling device) */
HYDRIUM_EXPORT int hydamd_multi_read(HydAmdMulti *m, uint8_t *dst, size_t capacity) {
    if (!m || !m->have_result)
        return m ? fail(m, HYD_API_ERROR, "no finished frame: hydamd_multi_result first", NULL) : HYD_API_ERROR;
    if (!dst || capacity < m->size)
        return fail(m, HYD_NEED_MORE_OUTPUT, "output buffer too small", NULL);
    const int st = hydamd_assembler_read(hydamd_context_assembler(m->ctx[m->assembling]), dst, m->size);
    return st ? fail(m, st, "read-back", m->ctx[m->assembling]) : HYD_OK;
}
