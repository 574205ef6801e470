/*
 * encoder.c — HYDEncoder session, the nine public hyd_* entry points and the host tile scheduler.
 *
 * This is the drop-in boundary (include/libhydrium/libhydrium.h).  Behaviour, argument meaning,
 * status codes and error strings follow the reference's libhydrium.c / encoder.c; what is behind
 * the boundary is new:
 *
 *   hyd_send_tile (reference libhydrium.c:172-203)
 *     -> geometry + "last tile" decision              (encoder.c:437-508 restated in tile_geometry)
 *     -> hydamd_encode_lf_group_host: the caller's samples are gathered into pinned staging and
 *        uploaded; the call returns once they have been read (the CLI reuses its row buffer right
 *        away, reference hydrium.c:416)
 *     -> on the frame's final tile only: hydamd_finish_frame runs the whole hot path for every
 *        LF group of the frame in batched launches (transform+tokenise, ANS tables, rANS, packing),
 *        then the host wraps the HF sections: LFGlobal, LF groups, HFGlobal, TOC (frame.c).
 *
 * One-frame mode therefore keeps the GPU fed with uploads while tiles arrive and codes the frame
 * in one go at the end, exactly when the reference, too, first produces frame bytes
 * (libhydrium.c:148-149).  There is NO CPU fallback: without a usable HIP device hyd_send_tile
 * fails with HYD_INTERNAL_ERROR.
 *
 * Deliberate deviations from the reference (all on inputs where the reference misbehaves):
 *   - frames with 128 or more than 255 LF groups are rejected (the reference never returns:
 *     uint8_t loop counter vs 256 clusters, entropy.c:99);
 *   - one-frame images must have every tile sent (the reference reads uninitialised state for
 *     unsent tiles, encoder.c:248-256,973-975);
 *   - non-finite float samples yield HYD_API_ERROR "Invalid NaN Float" (the reference sets the
 *     message but drops the status, format.c:172-174) — reported when the frame is finished;
 *   - output never lands in caller memory beyond the provided length (bitwriter.c:42-51 would
 *     realloc() the caller's buffer);
 *   - hyd_send_tile returns HYD_NEED_MORE_OUTPUT when coded bytes are still pending, as the public
 *     header documents (libhydrium.h:222-226); the reference's hyd_send_tile drops the status of its
 *     closing hyd_flush and always says HYD_OK (libhydrium.c:193-202), so a caller written to the
 *     documented protocol would truncate its file there;
 *   - tile-mode frames are pipelined: a tile's bytes (and an error its pixels cause) may surface up to
 *     seven calls later than the reference's, in send order; the final tile's call completes them all
 *     (tile_pipeline_depth below; the reference's own CLI loop, src/hydrium.c:463-476, is indifferent);
 *   - a one-frame tile sent twice is rejected with HYD_API_ERROR (the reference codes a corrupt
 *     frame: both copies land in the TOC permutation, encoder.c:241-325);
 *   - ICC profiles with the 'SGI ' or 'SUNW' platform signature: position 41 takes the default
 *     prediction, as a decoder must (it learns byte 41 only there); the reference indexes
 *     "I "[i - 42] with i = 41 (libhydrium.c:219-224), 4 GB past the literal, and crashes.
 */
#define _POSIX_C_SOURCE 200809L /* clock_gettime under -std=c99 */
#include <malloc.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "bitio.h"
#include "frame.h"
#include "prefix.h"
#include "hydrium_amd.h"
#include "libhydrium/libhydrium.h"

#define TILE_PIPE_MAX 8
#define HYD_MAX_DEVICES HYDAMD_MAX_PEERS /* devices one encoder can deal a frame to */

typedef struct LfgResult {
    int32_t *dc; /* [3][vbh][vbw]; NULL when the LF coefficients were coded on the device */
    uint8_t *lf_bits;                    /* device-coded LF-coefficient symbols (borrowed) or NULL */
    uint8_t lf_lengths[HYD_LF_CODES];
    uint32_t lf_alphabet, lf_run_pairs, lf_bit_count;
    uint32_t freq[HYD_FRAME_MAX_CLUSTERS][HYD_FRAME_ALPHABET];
    uint32_t alphabet[HYD_FRAME_MAX_CLUSTERS];
    uint32_t bits[HYDAMD_GROUPS_PER_LFG];
} LfgResult;

struct HYDEncoder {
    HYDImageMetadata metadata;
    int have_metadata;
    int one_frame;
    int level10;
    size_t lfg_count_x, lfg_count_y, lfg_per_frame;
    size_t tile_w, tile_h; /* pixels */
    const char *error;

    uint8_t *out; /* buffer on loan from the caller */
    size_t out_len, out_pos;

    HydBits stream;    /* codestream bytes produced and not yet handed over start at stream_pos */
    size_t stream_pos;
    int wrote_header;
    int last_tile;
    int frame_done; /* one-frame: the final tile has been coded */
    size_t tiles_sent;

    uint8_t *icc; /* mangled profile (libhydrium.c:242-305) */
    size_t icc_size;

    HydFrameLfg *sent; /* [lfg_per_frame] in send order */
    uint8_t *sent_mask; /* [lfg_per_frame] by raster id: one-frame tiles already received */
    HydAmdContext *dev;
    size_t dev_slots;  /* shape the context was created for */
    int dev_linear;
    int dev_failed;    /* a device call failed: do not park this context for reuse */

    /* tile mode: frames in flight (see tile_pipeline_depth).  A ring entry owns a device context for the life of the
     * encoder; e->dev is then only the entry being worked on */
    struct PendingTile {
        HydAmdContext *dev;
        int failed;
        int active;          /* launched, not yet collected */
        HydFrameLfg lfg;     /* the tile (the frame's only LF group) */
        HydFrameShape shape; /* shape.lfg points at lfg above */
    } pipe[TILE_PIPE_MAX];
    /* which device this encoder's single-device work runs on (taken in turn from the device list at the first tile), and,
     * for a one-frame image dealt to several devices: shard d owns the tiles sent first_slot[d] .. first_slot[d + 1] - 1 */
    int home_device;
    int home_index; /* its place in the device list: a sharded frame's shard d runs on list entry (home_index + d) mod count */
    int have_home;
    int shards; /* 0: not decided yet; 1: everything on e->dev; > 1: multi[] */
    struct Shard {
        HydAmdContext *dev;
        size_t first_slot, slots;
        int failed;
        int entry; /* place in the device list */
    } multi[HYD_MAX_DEVICES];
    int pipe_depth; /* 0: tile frames are coded synchronously through e->dev */
    int pipe_request; /* hydamd_set_tile_pipeline: frames in flight this encoder asks for; 0 = HYDAMD_TILE_PIPELINE, else 1 */
    struct PendingTile *cur_pend; /* the ring entry the running call works for (device errors are recorded there) */
    size_t tile_seq;
    char verify_msg[128]; /* HYDAMD_VERIFY_PEERS: the error text that names the device pair */
};

#define FAIL(enc, code, msg) ((enc)->error = (msg), (code))

/* HYDAMD_TRACE=1 prints where the host-pointer API path spends its wall time (stderr) */
static double now_ms(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
static int trace_on(void) {
    static int on = -1;
    if (on < 0) {
        const char *e = getenv("HYDAMD_TRACE");
        on = e && *e && *e != '0';
    }
    return on;
}
#define TRACE(label, t0)                                                          \
    do {                                                                          \
        if (trace_on())                                                           \
            fprintf(stderr, "[hydrium] %-28s %8.3f ms\n", (label), now_ms() - (t0)); \
    } while (0)

/* ---------------------------------------------------------------------------------------------
 * frame assembly (shared by the product path and the CPU-only test hook)
 * ------------------------------------------------------------------------------------------- */

/* The LF groups of a frame are independent prefix-coded sections; in a frame with more than one
 * group each starts on a byte boundary, so they are coded by a few host threads into private
 * buffers and appended in send order. */
typedef struct LfWork {
    const HydFrameShape *shape;
    const LfgResult *res;
    HydBits *out;       /* [lfg_count] */
    int *status;        /* [lfg_count] */
    const char **err;   /* [lfg_count] */
    size_t first, stride;
} LfWork;

/* The HF-metadata sub-streams of an LF group depend on its geometry only (frame.c:
 * hyd_write_lf_group_tail), cost ~200k symbol sends for a full LF group and compress to a few
 * hundred bytes: keep them per (vbw, vbh) for the life of the process.  Entries are immutable once
 * published, so readers only need the lock to find them. */
typedef struct TailEntry {
    size_t vbw, vbh;
    HydBits bits;
} TailEntry;
static pthread_mutex_t g_tail_lock = PTHREAD_MUTEX_INITIALIZER;
static TailEntry g_tails[32];
static int g_ntails;

static const HydBits *find_tail_locked(size_t vbw, size_t vbh) {
    for (int i = 0; i < g_ntails; i++)
        if (g_tails[i].vbw == vbw && g_tails[i].vbh == vbh)
            return &g_tails[i].bits;
    return NULL;
}

static const HydBits *lf_tail(size_t vbw, size_t vbh) {
    pthread_mutex_lock(&g_tail_lock);
    const HydBits *hit = find_tail_locked(vbw, vbh);
    pthread_mutex_unlock(&g_tail_lock);
    if (hit)
        return hit;
    HydBits fresh;
    const char *err = NULL;
    hb_init(&fresh);
    if (hyd_write_lf_group_tail(&fresh, vbw, vbh, &err) || fresh.failed) {
        hb_free(&fresh);
        return NULL; /* the caller codes it inline and reports the error there */
    }
    pthread_mutex_lock(&g_tail_lock);
    hit = find_tail_locked(vbw, vbh);
    if (!hit && g_ntails < (int)(sizeof(g_tails) / sizeof(g_tails[0]))) {
        g_tails[g_ntails].vbw = vbw;
        g_tails[g_ntails].vbh = vbh;
        g_tails[g_ntails].bits = fresh;
        hit = &g_tails[g_ntails++].bits;
        fresh.data = NULL;
    }
    pthread_mutex_unlock(&g_tail_lock);
    if (fresh.data)
        hb_free(&fresh);
    return hit;
}

static int write_one_lf_group(HydBits *out, const LfgResult *r, size_t vbw, size_t vbh, const char **err) {
    if (r->lf_bits) {
        const HydLfCoded lf = {r->lf_lengths, r->lf_alphabet, r->lf_run_pairs, r->lf_bits, r->lf_bit_count};
        return hyd_write_lf_group_coded(out, vbw, vbh, &lf, lf_tail(vbw, vbh), err);
    }
    return hyd_write_lf_group(out, r->dc, vbw, vbh, err);
}

static void *lf_worker(void *arg) {
    const LfWork *w = arg;
    for (size_t s = w->first; s < w->shape->lfg_count; s += w->stride) {
        const size_t vbw = (w->shape->lfg[s].width + 7) >> 3, vbh = (w->shape->lfg[s].height + 7) >> 3;
        hb_init(&w->out[s]);
        w->err[s] = NULL;
        w->status[s] = write_one_lf_group(&w->out[s], &w->res[s], vbw, vbh, &w->err[s]);
        hb_align(&w->out[s]);
    }
    return NULL;
}

static int code_lf_groups_parallel(HYDEncoder *e, const HydFrameShape *shape, const LfgResult *res, HydBits *out) {
    const size_t n = shape->lfg_count;
    int *status = calloc(n, sizeof(int));
    const char **err = calloc(n, sizeof(char *));
    if (!status || !err) {
        free(status);
        free(err);
        return FAIL(e, HYD_NOMEM, "out of memory");
    }
    long cores = sysconf(_SC_NPROCESSORS_ONLN);
    size_t threads = cores > 1 ? (size_t)cores : 1;
    if (threads > n)
        threads = n;
    if (threads > 16)
        threads = 16;
    LfWork work[16];
    pthread_t tid[16];
    size_t started = 0;
    for (size_t i = 0; i < threads; i++) {
        work[i] = (LfWork){shape, res, out, status, err, i, threads};
        if (i + 1 < threads && pthread_create(&tid[started], NULL, lf_worker, &work[i]) == 0)
            started++;
        else
            lf_worker(&work[i]); /* the calling thread takes the last share (and any that could not be spawned) */
    }
    for (size_t i = 0; i < started; i++)
        pthread_join(tid[i], NULL);
    int ret = 0;
    for (size_t s = 0; s < n && !ret; s++) {
        if (status[s] || out[s].failed) {
            e->error = err[s] ? err[s] : "LF group coding failed";
            ret = status[s] ? status[s] : HYD_NOMEM;
        }
    }
    free(status);
    free(err);
    return ret;
}

static int device_fail(HYDEncoder *e, int code);
static void pipe_release(HYDEncoder *e);
static void multi_release(HYDEncoder *e);

/* payload == NULL: the packed HF sections are still on the device (e->dev) and are copied straight
 * into the output stream */
typedef struct PayloadSegments { /* the packed HF sections in pieces (one per shard blob) instead of one string */
    size_t count;
    const uint8_t *const *ptr;
    const size_t *len;
} PayloadSegments;

static int assemble_frame(HYDEncoder *e, const HydFrameShape *shape, const LfgResult *res, unsigned max_alphabet,
                          const uint8_t *payload, size_t payload_len, HydBits *lf_prebuilt, const PayloadSegments *segs) {
    uint8_t *fetched = NULL;
    const size_t fg = ((shape->frame_width + 255) >> 8) * ((shape->frame_height + 255) >> 8);
    const int multi = fg > 1;
    const size_t toc_n = hyd_toc_entries(shape);
    const unsigned num_presets = (unsigned)shape->lfg_count;
    int ret = 0;
    HydBits body;
    hb_init(&body);
    size_t *sizes = calloc(toc_n, sizeof(size_t));
    uint32_t(*freq)[HYD_FRAME_MAX_CLUSTERS][HYD_FRAME_ALPHABET] = calloc(num_presets, sizeof(*freq));
    uint32_t(*alpha)[HYD_FRAME_MAX_CLUSTERS] = calloc(num_presets, sizeof(*alpha));
    if (!sizes || !freq || !alpha) {
        ret = FAIL(e, HYD_NOMEM, "out of memory");
        goto done;
    }
    size_t k = 0, mark = 0;
    int overflow = 0; /* more sections than the frame geometry has TOC entries: inconsistent LF-group list */
#define PUSH_SIZE(v)                       \
    do {                                   \
        if (k < toc_n)                     \
            sizes[k++] = (v);              \
        else                               \
            overflow = 1;                  \
    } while (0)
#define CLOSE_SECTION()                    \
    do {                                   \
        if (multi) {                       \
            hb_align(&body);               \
            PUSH_SIZE(body.len - mark);    \
            mark = body.len;               \
        }                                  \
    } while (0)

    double t0 = now_ms();
    hyd_write_lf_global(&body);
    CLOSE_SECTION();
    if (multi && lf_prebuilt) { /* coded while the GPU was still busy with the entropy stage (finish_frame) */
        for (size_t s = 0; s < shape->lfg_count; s++) {
            hb_append_bytes(&body, lf_prebuilt[s].data, lf_prebuilt[s].len);
            CLOSE_SECTION();
        }
    } else if (multi && shape->lfg_count > 1) {
        HydBits *lf = calloc(shape->lfg_count, sizeof(HydBits));
        if (!lf) {
            ret = FAIL(e, HYD_NOMEM, "out of memory");
            goto done;
        }
        ret = code_lf_groups_parallel(e, shape, res, lf);
        for (size_t s = 0; s < shape->lfg_count; s++) {
            if (!ret) {
                hb_append_bytes(&body, lf[s].data, lf[s].len);
                CLOSE_SECTION();
            }
            hb_free(&lf[s]);
        }
        free(lf);
        if (ret)
            goto done;
    } else {
        for (size_t s = 0; s < shape->lfg_count; s++) {
            const size_t vbw = (shape->lfg[s].width + 7) >> 3, vbh = (shape->lfg[s].height + 7) >> 3;
            ret = write_one_lf_group(&body, &res[s], vbw, vbh, &e->error);
            if (ret)
                goto done;
            CLOSE_SECTION();
        }
    }
    TRACE("  LF group sections", t0);
    t0 = now_ms();
    /* tables are signalled per preset = raster LF-group id, whatever the send order was */
    for (size_t s = 0; s < shape->lfg_count; s++) {
        const size_t p = shape->lfg[s].raster_id;
        memcpy(freq[p], res[s].freq, sizeof(res[s].freq));
        memcpy(alpha[p], res[s].alphabet, sizeof(res[s].alphabet));
    }
    ret = hyd_write_hf_global(&body, num_presets, fg, (const uint32_t(*)[HYD_FRAME_MAX_CLUSTERS][HYD_FRAME_ALPHABET])freq,
                              (const uint32_t(*)[HYD_FRAME_MAX_CLUSTERS])alpha, max_alphabet, &e->error);
    if (ret)
        goto done;
    CLOSE_SECTION();
    TRACE("  HFGlobal", t0);
    t0 = now_ms();
    if (multi) {
        /* the device payload already is: byte-padded sections, send order, raster inside an LF group;
         * it follows the body, so only its sizes are needed here */
        for (size_t s = 0; s < shape->lfg_count; s++) {
            const size_t ng = ((shape->lfg[s].width + 255) >> 8) * ((shape->lfg[s].height + 255) >> 8);
            for (size_t g = 0; g < ng && g < HYDAMD_GROUPS_PER_LFG; g++)
                PUSH_SIZE((res[s].bits[g] + 7u) >> 3);
        }
    } else {
        /* a single-group frame is one bit-contiguous section (encoder.c:837-850,968-981 guards) */
        if (!payload && payload_len) {
            fetched = malloc(payload_len);
            if (!fetched) {
                ret = FAIL(e, HYD_NOMEM, "out of memory");
                goto done;
            }
            ret = hydamd_read_payload(e->dev, fetched, payload_len);
            if (ret) {
                ret = device_fail(e, ret);
                goto done;
            }
            payload = fetched;
        }
        hb_append_bits(&body, payload, res[0].bits[0]);
    }
    hb_align(&body);
    if (!multi)
        PUSH_SIZE(body.len);
    if (overflow || k != toc_n || body.failed) {
        ret = FAIL(e, body.failed ? HYD_NOMEM : HYD_INTERNAL_ERROR, "frame assembly inconsistency");
        goto done;
    }
    ret = hyd_write_frame_header(&e->stream, shape, &e->error);
    if (!ret)
        ret = hyd_write_toc_sizes(&e->stream, sizes, toc_n);
    if (ret) {
        if (!e->error)
            e->error = "frame header could not be written";
        goto done;
    }
    hb_append_bytes(&e->stream, body.data, body.len);
    if (multi && payload_len) {
        if (segs) { /* straight from the shards' blobs into the output: the only copy the sections see here */
            uint8_t *dst = hb_extend(&e->stream, payload_len);
            for (size_t i = 0; dst && i < segs->count; i++) {
                memcpy(dst, segs->ptr[i], segs->len[i]);
                dst += segs->len[i];
            }
        } else if (payload) {
            hb_append_bytes(&e->stream, payload, payload_len);
        } else {
            uint8_t *dst = hb_extend(&e->stream, payload_len);
            if (dst && (ret = hydamd_read_payload(e->dev, dst, payload_len)) != 0)
                ret = device_fail(e, ret);
        }
    }
    if (!ret && e->stream.failed)
        ret = FAIL(e, HYD_NOMEM, "out of memory");
    TRACE("  frame header, TOC, body + HF sections", t0);
done:
    free(fetched);
#undef CLOSE_SECTION
#undef PUSH_SIZE
    free(sizes);
    free(freq);
    free(alpha);
    hb_free(&body);
    return ret;
}

static void take_spare_buffer(HydBits *b); /* below, with the other spare-buffer functions */
static void take_spare_buffer_for(HydBits *b, size_t want);

static int emit_file_header(HYDEncoder *e) {
    if (e->wrote_header)
        return 0;
    /* a byte per four pixels is more than photographic content needs; an encoder for a small image does not
     * walk off with the 256 MB buffer a 16K frame left behind */
    take_spare_buffer_for(&e->stream, (size_t)e->metadata.width * e->metadata.height / 4 + 65536);
    int ret = hyd_write_file_header(&e->stream, e->metadata.width, e->metadata.height, e->level10, e->icc, e->icc_size,
                                    &e->error);
    if (ret) {
        if (!e->error)
            e->error = "file header could not be written";
        return ret;
    }
    e->wrote_header = 1;
    return 0;
}

/* move pending codestream bytes into the caller's buffer */
static void drain(HYDEncoder *e) {
    if (!e->out)
        return;
    size_t n = e->stream.len - e->stream_pos;
    if (n > e->out_len - e->out_pos)
        n = e->out_len - e->out_pos;
    if (n) {
        hydamd_host_copy(e->out + e->out_pos, e->stream.data + e->stream_pos, n);
        e->out_pos += n;
        e->stream_pos += n;
    }
    if (e->stream_pos == e->stream.len) {
        hb_reset(&e->stream);
        e->stream_pos = 0;
    }
}

/* ---------------------------------------------------------------------------------------------
 * device context reuse
 *
 * Creating a context allocates its buffers (about 50 MB per LF-group slot, more once a frame has
 * outgrown them) and pinned staging, and destroying it gives them back: several milliseconds per
 * image together.  Idle contexts are therefore parked when their encoder is destroyed and handed to
 * the next encoder that needs the same shape (a few of them, so that several threads encoding images
 * back to back each find one).  Contexts that saw a device error are not parked, nor is one that would
 * take the parked total over HYDAMD_CONTEXT_CACHE_MB (default 8192 MB of device memory, counted at 50 MB
 * per slot + staging); HYDAMD_CONTEXT_CACHE=0 turns the parking off and hydamd_trim_cache() empties it.
 * ------------------------------------------------------------------------------------------- */
#define CTX_POOL_MAX 32
typedef struct ParkedCtx {
    HydAmdContext *ctx;
    size_t slots;
    int linear;
    unsigned long stamp; /* larger = parked more recently */
} ParkedCtx;
static pthread_mutex_t g_ctx_lock = PTHREAD_MUTEX_INITIALIZER;
static ParkedCtx g_pool[CTX_POOL_MAX];
static unsigned long g_stamp;

/* HYDAMD_CONTEXT_CACHE: how many idle contexts may stay parked (default 16, at most 32, 0 = none): one per
 * thread that encodes images back to back is what a batch job wants, one per frame in flight a tile-mode encoder */
static int ctx_pool_size(void) {
    static int n = -1;
    if (n < 0) {
        const char *v = getenv("HYDAMD_CONTEXT_CACHE");
        n = v && *v ? atoi(v) : 16;
        if (n < 0)
            n = 0;
        if (n > CTX_POOL_MAX)
            n = CTX_POOL_MAX;
    }
    return n;
}

/* HYDAMD_EAGER=0: run the transform kernels when the frame's final tile arrives instead of tile by tile (A/B) */
static int eager_on(void) {
    static int on = -1;
    if (on < 0) {
        const char *v = getenv("HYDAMD_EAGER");
        on = !(v && *v == '0');
    }
    return on;
}

/* Tile mode (tile_size_shift >= 0) makes every tile a frame of its own, and the reference hands its bytes over before
 * hyd_send_tile returns (encoder.c:339-378, 1008).  That is the DEFAULT here too (depth 1): a call's frame is complete
 * when the call returns, a device or NaN error is reported by the call that sent the offending tile, and an encoder
 * abandoned mid-image has emitted every frame the reference would have.  Done that way on a GPU a tile costs the latency
 * of the whole kernel sequence — 2-3 ms, most of it the serial rANS chain of its longest group, whether the tile is
 * 256x256 or 2048x2048 (measured: 32 Mpixel/s for 256x256 tiles, no better than one CPU core).  A caller that wants the
 * throughput instead asks for it: hydamd_set_tile_pipeline(encoder, depth) (include/hydrium_amd.h) or the environment's
 * HYDAMD_TILE_PIPELINE=depth (2..8) keeps up to `depth` tile frames in flight, each on a device context of its own: a
 * call stages and launches its tile and collects the frame launched `depth` calls earlier; frames reach the output in
 * send order, and the call that sends the image's final tile collects everything.  What such a caller sees: a tile's bytes
 * (and a NaN or device error) arrive up to depth - 1 calls later than the reference's; the documented protocol — write
 * what hyd_release_output_buffer reports, loop on HYD_NEED_MORE_OUTPUT — copes (in one-frame mode every call but the
 * last already yields nothing); an encoder destroyed before its final tile drops the frames still in flight. */
static int tile_pipeline_depth(const HYDEncoder *e) {
    static int env = -1;
    if (env < 0) {
        const char *v = getenv("HYDAMD_TILE_PIPELINE");
        env = v && *v ? atoi(v) : 1;
        if (env < 1)
            env = 1;
        if (env > TILE_PIPE_MAX)
            env = TILE_PIPE_MAX;
    }
    return e->pipe_request ? e->pipe_request : env;
}

/* The devices the drop-in API encodes on (the host tile scheduler's view of the node):
 *   HYDAMD_DEVICES=a,b,c   this list, in this order (an index may repeat: several contexts of one GPU — how the tests
 *                          run the multi-device path on a box with one);
 *   HYDAMD_DEVICE=n        one device (a process per GPU sets its own; what bench.py's ranks do);
 *   neither                every device the runtime shows, at most HYD_MAX_DEVICES.
 * A one-frame image of at least HYDAMD_SHARD_MIN_LF_GROUPS (default 8) LF groups is dealt to the devices in runs of
 * consecutive tiles, two or more per device (multi_* below); anything smaller — and every tile-mode encoder — runs on ONE
 * device, encoders taking the devices in turn (a batch of frames on several threads spreads over the node). */
static int g_devices[HYD_MAX_DEVICES], g_device_count = -1;
static unsigned long g_encoder_seq; /* under g_ctx_lock */
static int g_shard_min = 8;

static void device_list_init(void) {
    pthread_mutex_lock(&g_ctx_lock);
    if (g_device_count < 0) {
        int n = 0;
        const char *v = getenv("HYDAMD_DEVICES");
        if (v && *v) {
            while (*v && n < HYD_MAX_DEVICES) {
                char *end = NULL;
                const long d = strtol(v, &end, 10);
                if (end == v)
                    break;
                g_devices[n++] = d < 0 ? 0 : (int)d;
                v = *end == ',' ? end + 1 : end;
            }
        } else if ((v = getenv("HYDAMD_DEVICE")) != NULL && *v) {
            g_devices[n++] = atoi(v) < 0 ? 0 : atoi(v);
        } else {
            n = hydamd_device_count();
            if (n > HYD_MAX_DEVICES)
                n = HYD_MAX_DEVICES;
            for (int i = 0; i < n; i++)
                g_devices[i] = i;
        }
        if (n < 1) { /* no device: hydamd_create says so when a tile arrives */
            g_devices[0] = 0;
            n = 1;
        }
        if ((v = getenv("HYDAMD_SHARD_MIN_LF_GROUPS")) != NULL && *v)
            g_shard_min = atoi(v) < 2 ? 2 : atoi(v);
        g_device_count = n;
    }
    pthread_mutex_unlock(&g_ctx_lock);
}

static HydAmdContext *ctx_acquire(int device, size_t slots, int linear, int *status) {
    HydAmdContext *c = NULL;
    pthread_mutex_lock(&g_ctx_lock);
    for (int i = 0; i < CTX_POOL_MAX && !c; i++)
        if (g_pool[i].ctx && g_pool[i].slots == slots && g_pool[i].linear == linear && hydamd_context_device(g_pool[i].ctx) == device) {
            c = g_pool[i].ctx;
            g_pool[i].ctx = NULL;
        }
    pthread_mutex_unlock(&g_ctx_lock);
    if (c) {
        (void)hydamd_forget_content(c); /* the density of whatever image it coded last says nothing about the next one */
        *status = HYD_OK;
        return c;
    }
    return hydamd_create(device, (int)slots, linear, 0, status);
}

/* what a parked context of `slots` LF-group slots holds, in MB: its frame arrays at their default sizes,
 * the LF coder's, one staging tile per slot for 16-bit samples (hydamd_token_capacity tells whether it grew) */
static size_t ctx_megabytes(HydAmdContext *c, size_t slots) {
    const size_t per_slot = hydamd_token_capacity(c) > 98304 ? 200 : 75;
    return slots * per_slot + 64;
}

static size_t ctx_pool_megabytes(void) {
    static long mb = -1;
    if (mb < 0) {
        const char *v = getenv("HYDAMD_CONTEXT_CACHE_MB");
        mb = v && *v ? atol(v) : 8192;
        if (mb < 0)
            mb = 0;
    }
    return (size_t)mb;
}

static void ctx_release(HydAmdContext *c, size_t slots, int linear, int healthy) {
    const int cap = ctx_pool_size();
    /* the wait for the context's stream happens before the lock; the budget check and the insertion under
     * ONE acquisition of it, so that two encoders destroyed at the same time cannot both pass the check */
    if (healthy && cap > 0 && hydamd_sync(c) == HYD_OK) {
        const size_t mine = ctx_megabytes(c, slots);
        pthread_mutex_lock(&g_ctx_lock);
        int where = -1;
        for (int i = 0; i < cap && where < 0; i++)
            if (!g_pool[i].ctx)
                where = i;
        if (where < 0) { /* full: the context parked longest ago makes room */
            where = 0;
            for (int i = 1; i < cap; i++)
                if (g_pool[i].stamp < g_pool[where].stamp)
                    where = i;
        }
        size_t parked_mb = 0;
        for (int i = 0; i < CTX_POOL_MAX; i++)
            if (g_pool[i].ctx && i != where)
                parked_mb += ctx_megabytes(g_pool[i].ctx, g_pool[i].slots);
        if (parked_mb + mine <= ctx_pool_megabytes()) {
            HydAmdContext *evicted = g_pool[where].ctx;
            g_pool[where] = (ParkedCtx){c, slots, linear, ++g_stamp};
            c = evicted;
        }
        pthread_mutex_unlock(&g_ctx_lock);
    }
    if (c)
        hydamd_destroy(c);
}

/* ---------------------------------------------------------------------------------------------
 * public API
 * ------------------------------------------------------------------------------------------- */

/* Frames in flight (tile-mode pipelining, encoders on several threads) live on HIP streams of their own, and the runtime
 * spreads streams over GPU_MAX_HW_QUEUES hardware queues — 4 unless the process says otherwise, which serialises what was
 * meant to overlap (eight tile frames in flight: 1.47 ms per 256x256 tile with 4 queues, 0.57 with 20; a batch of 4K
 * frames on 8 threads: 628 frames/s against 1037).  That is the DEPLOYMENT's setting to make (export GPU_MAX_HW_QUEUES=20
 * before the process starts): a drop-in library does not edit its host's environment (until
 * round 3 a load-time constructor did — it changed HIP for every other user of the process and raced getenv in threads). */

HYDRIUM_EXPORT HYDEncoder *hyd_encoder_new(void) {
    HYDEncoder *e = calloc(1, sizeof(HYDEncoder));
    if (e)
        hb_init(&e->stream);
    return e;
}

/* ---------------------------------------------------------------------------------------------
 * Spare stream buffers.  A frame is assembled in a growable buffer of its size (12 MB for 8K, 50 MB for
 * 16K); fresh from malloc that is 2-8 ms of page faults per frame.  Buffers of destroyed encoders, and
 * frame buffers that come back through hydamd_free, are kept (mapped) for the next frame instead: at most
 * SPARE_SLOTS of them, each between 1 MB and 256 MB; hydamd_trim_cache() drops them.
 * ------------------------------------------------------------------------------------------- */
#define SPARE_SLOTS 4
#define SPARE_MIN ((size_t)1 << 20)
#define SPARE_MAX ((size_t)1 << 28)
typedef struct SpareBuf {
    void *p;
    size_t cap;
} SpareBuf;
static pthread_mutex_t g_buf_lock = PTHREAD_MUTEX_INITIALIZER;
static SpareBuf g_spare[SPARE_SLOTS];

/* an empty stream takes a spare buffer: the smallest one that holds `want` bytes (0: unknown), else the largest */
static void take_spare_buffer_for(HydBits *b, size_t want) {
    if (b->data)
        return;
    pthread_mutex_lock(&g_buf_lock);
    int best = -1;
    for (int i = 0; i < SPARE_SLOTS; i++) {
        if (!g_spare[i].p)
            continue;
        if (best < 0) {
            best = i;
            continue;
        }
        const int fits = want && g_spare[i].cap >= want, best_fits = want && g_spare[best].cap >= want;
        if (fits ? (!best_fits || g_spare[i].cap < g_spare[best].cap) : (!best_fits && g_spare[i].cap > g_spare[best].cap))
            best = i;
    }
    if (best >= 0) {
        b->data = g_spare[best].p;
        b->cap = g_spare[best].cap;
        g_spare[best].p = NULL;
        g_spare[best].cap = 0;
    }
    pthread_mutex_unlock(&g_buf_lock);
}
static void take_spare_buffer(HydBits *b) { take_spare_buffer_for(b, 0); }

/* keeps p (returns 1) in a free slot or in place of a smaller spare, which is freed */
static int offer_spare_buffer(void *p, size_t cap) {
    if (!p || cap < SPARE_MIN || cap > SPARE_MAX)
        return 0;
    void *drop = NULL;
    int kept = 0, smallest = 0;
    pthread_mutex_lock(&g_buf_lock);
    for (int i = 0; i < SPARE_SLOTS && !kept; i++) {
        if (!g_spare[i].p) {
            g_spare[i].p = p;
            g_spare[i].cap = cap;
            kept = 1;
        } else if (g_spare[i].cap < g_spare[smallest].cap) {
            smallest = i;
        }
    }
    if (!kept && g_spare[smallest].cap < cap) {
        drop = g_spare[smallest].p;
        g_spare[smallest].p = p;
        g_spare[smallest].cap = cap;
        kept = 1;
    }
    pthread_mutex_unlock(&g_buf_lock);
    free(drop);
    return kept;
}

HYDRIUM_EXPORT HYDStatusCode hyd_encoder_destroy(HYDEncoder *e) {
    if (!e)
        return HYD_OK;
    multi_release(e);
    pipe_release(e);
    if (e->dev) {
        const double t0 = now_ms();
        ctx_release(e->dev, e->dev_slots, e->dev_linear, !e->dev_failed);
        TRACE("release device context", t0);
    }
    if (offer_spare_buffer(e->stream.data, e->stream.cap))
        e->stream.data = NULL;
    hb_free(&e->stream);
    free(e->icc);
    free(e->sent);
    free(e->sent_mask);
    free(e);
    return HYD_OK;
}

HYDRIUM_EXPORT const char *hyd_error_message_get(HYDEncoder *e) { return e->error; }

HYDRIUM_EXPORT HYDStatusCode hyd_set_metadata(HYDEncoder *e, const HYDImageMetadata *md) {
    /* validation order and messages: libhydrium.c:46-78 */
    if (!md->width || !md->height)
        return FAIL(e, HYD_API_ERROR, "invalid zero-width or zero-height");
    const uint64_t w = md->width, h = md->height;
    if (w > UINT64_C(1) << 30 || h > UINT64_C(1) << 30)
        return FAIL(e, HYD_API_ERROR, "width or height out of bounds");
    if (w * h > UINT64_C(1) << 40)
        return FAIL(e, HYD_API_ERROR, "width times height out of bounds");
    e->metadata = *md;
    if (w > (1 << 20) || h > (1 << 20) || w * h > (1 << 28))
        e->level10 = 1;
    if (md->tile_size_shift_x < -1 || md->tile_size_shift_x > 3)
        return FAIL(e, HYD_API_ERROR, "tile_size_shift_y must be between -1 and 3"); /* sic, libhydrium.c:70-72 */
    if (md->tile_size_shift_y < -1 || md->tile_size_shift_y > 3)
        return FAIL(e, HYD_API_ERROR, "tile_size_shift_y must be between -1 and 3");
    e->one_frame = md->tile_size_shift_x < 0 || md->tile_size_shift_y < 0;
    e->lfg_count_y = (md->height + 2047) >> 11;
    e->lfg_count_x = (md->width + 2047) >> 11;
    e->lfg_per_frame = e->one_frame ? e->lfg_count_x * e->lfg_count_y : 1;
    e->tile_w = e->one_frame ? 2048 : (size_t)256 << md->tile_size_shift_x;
    e->tile_h = e->one_frame ? 2048 : (size_t)256 << md->tile_size_shift_y;
    if (e->lfg_per_frame > HYDAMD_MAX_LF_GROUPS || e->lfg_per_frame == 128)
        return FAIL(e, HYD_API_ERROR, "one frame cannot hold 128 or more than 255 LF groups; use tile mode");
    free(e->sent);
    free(e->sent_mask);
    e->sent = calloc(e->lfg_per_frame, sizeof(HydFrameLfg));
    e->sent_mask = calloc(e->lfg_per_frame, 1);
    if (!e->sent || !e->sent_mask)
        return FAIL(e, HYD_NOMEM, "out of memory");
    multi_release(e);
    pipe_release(e);
    if (e->dev) { /* metadata changed: the device context is rebuilt lazily */
        ctx_release(e->dev, e->dev_slots, e->dev_linear, !e->dev_failed);
        e->dev = NULL;
    }
    e->have_metadata = 1;
    e->tile_seq = 0;
    e->tiles_sent = 0;
    e->frame_done = 0;
    return HYD_OK;
}

HYDRIUM_EXPORT HYDStatusCode hyd_provide_output_buffer(HYDEncoder *e, uint8_t *buffer, size_t buffer_len) {
    if (buffer_len < 64)
        return FAIL(e, HYD_API_ERROR, "provided buffer must be at least 64 bytes long");
    if (e->out)
        return FAIL(e, HYD_API_ERROR, "buffer was already provided");
    if (!buffer)
        return FAIL(e, HYD_API_ERROR, "buffer may not be null");
    e->out = buffer;
    e->out_len = buffer_len;
    e->out_pos = 0;
    return HYD_OK;
}

HYDRIUM_EXPORT HYDStatusCode hyd_release_output_buffer(HYDEncoder *e, size_t *written) {
    if (!e->out)
        return FAIL(e, HYD_API_ERROR, "buffer was never provided");
    *written = e->out_pos;
    e->out = NULL;
    return HYD_OK;
}

HYDRIUM_EXPORT HYDStatusCode hyd_flush(HYDEncoder *e) {
    if (e->one_frame && !e->last_tile)
        return HYD_OK; /* libhydrium.c:148-149 */
    if (!e->out)
        return FAIL(e, HYD_API_ERROR, "buffer was never provided");
    drain(e);
    return e->stream_pos < e->stream.len ? HYD_NEED_MORE_OUTPUT : HYD_OK;
}

/* the context e->dev is not to be parked for reuse; in tile mode that is recorded in the ring entry that owns it,
 * whatever the call does next (e->dev_failed alone is overwritten when the next call switches to another entry) */
static void mark_device_failed(HYDEncoder *e) {
    e->dev_failed = 1;
    if (e->cur_pend)
        e->cur_pend->failed = 1;
    for (int d = 0; d < e->shards && e->shards > 1; d++)
        if (e->multi[d].dev == e->dev)
            e->multi[d].failed = 1;
}

static int device_fail(HYDEncoder *e, int code) {
    mark_device_failed(e);
    /* hydamd status codes are HYDStatusCode values; keep a static string for the message */
    static const char *const generic = "GPU encode failed (see hydamd_error)";
    const char *m = hydamd_error(e->dev);
    if (m && getenv("HYDAMD_TRACE"))
        fprintf(stderr, "[hydrium] device error %d: %s\n", code, m);
    if (m && strstr(m, "NaN"))
        e->error = "Invalid NaN Float";
    else if (m && strstr(m, "no usable HIP device"))
        e->error = "no usable HIP device (this build has no CPU fallback)";
    else if (code == HYD_NOMEM)
        e->error = "out of device memory";
    else
        e->error = generic;
    return code;
}

/* HYDAMD_HOST_ASSEMBLY=1: build frames on the host from read-back results, as round 2 did (A/B measurements, and the
 * path every frame of a single group and every tile-mode frame still takes) */
static int host_assembly_forced(void) {
    static int on = -1;
    if (on < 0) {
        const char *v = getenv("HYDAMD_HOST_ASSEMBLY");
        on = v && *v && *v != '0';
    }
    return on;
}

/* The frame put together on the GPU (csrc/hip/assemble.hip): the context exports its results as one blob in
 * device memory, the assembler's kernels write every section into place behind the entropy stage, and the
 * finished frame comes back in ONE copy — where the host used to read back tables, section sizes and LF
 * streams slot by slot, code the LF group sections itself and splice everything together. */
static int finish_frame_on_device(HYDEncoder *e, const HydFrameShape *shape) {
    const size_t n = shape->lfg_count;
    double t0 = now_ms();
    int ret = hydamd_finish_frame(e->dev, (int)n);
    if (ret)
        return device_fail(e, ret);
    HydAmdAssembler *as = hydamd_context_assembler(e->dev);
    if (!as)
        return device_fail(e, HYD_INTERNAL_ERROR);
    uint32_t lf_ids[HYDAMD_MAX_LF_GROUPS];
    for (size_t s = 0; s < n; s++)
        lf_ids[s] = (uint32_t)shape->lfg[s].raster_id;
    const uint32_t slots = (uint32_t)n;
    ret = hydamd_assembler_plan(as, &e->metadata, 0, 1, 1, &slots, lf_ids, NULL, 0);
    if (ret)
        return FAIL(e, ret, "frame description rejected by the assembler");
    /* the blob is a view: the output's size comes from the context's capacities plus what the assembler itself adds per
     * LF group (head bits, TOC entries) and per frame (prefix, HFGlobal); should a frame still exceed it, the assembler
     * says how many bytes it needs and the frame is assembled again into a buffer of that size */
    size_t out_cap = hydamd_blob_bound(e->dev, (int)n) + 4096 * n + (256u << 10);
    for (int attempt = 0; attempt < 4; attempt++) {
        const void *blob = NULL;
        size_t cap = 0, size = 0;
        ret = hydamd_export_frame_owned(e->dev, (int)n, &blob, &cap);
        if (!ret)
            ret = hydamd_assembler_run(as, &blob, &cap, hydamd_get_stream(e->dev), NULL, out_cap);
        if (ret)
            return device_fail(e, ret);
        ret = hydamd_sync(e->dev); /* a frame that outgrew the context's buffers is rerun in here: its blob is then stale */
        if (ret)
            return device_fail(e, ret);
        TRACE("GPU hot path + assembly", t0);
        t0 = now_ms();
        ret = hydamd_assembler_result(as, &size);
        if (ret) {
            const char *m = hydamd_assembler_error(as);
            if (m && strstr(m, "incomplete"))
                continue; /* export the rerun frame's results and assemble again */
            if (ret == HYD_NEED_MORE_OUTPUT && size > out_cap) {
                out_cap = size;
                continue;
            }
            if (m && strstr(m, "NaN"))
                return FAIL(e, HYD_API_ERROR, "Invalid NaN Float");
            mark_device_failed(e);
            return FAIL(e, ret < HYD_ERROR_START ? ret : HYD_INTERNAL_ERROR, "GPU frame assembly failed");
        }
        uint8_t *dst = hb_extend(&e->stream, size);
        if (!dst)
            return FAIL(e, HYD_NOMEM, "out of memory");
        ret = hydamd_assembler_read(as, dst, size);
        TRACE("frame to the host", t0);
        return ret ? device_fail(e, ret) : 0;
    }
    mark_device_failed(e);
    return FAIL(e, HYD_INTERNAL_ERROR, "frame still does not fit after enlarging its buffers");
}

/* frames assembled on the host from ONE read-back (hydamd_stage_frame_blob / hydamd_read_frame_blob) instead of one small
 * copy per table, size array and byte string: frames of a single LF group with the LF coder on the device — every tile-mode
 * frame.  HYDAMD_STAGED_READBACK=0 keeps the separate copies (A/B) */
static int staged_readback(const HYDEncoder *e, size_t n) {
    static int on = -1;
    if (on < 0) {
        const char *v = getenv("HYDAMD_STAGED_READBACK");
        on = !(v && *v == '0');
    }
    return on && n == 1 && hydamd_lf_coder(e->dev);
}

/* The host-assembly path in two halves, both on e->dev: every kernel of the frame is enqueued (nothing waits) ... */
static int finish_frame_launch(HYDEncoder *e, const HydFrameShape *shape) {
    const size_t n = shape->lfg_count;
    const size_t fg = ((shape->frame_width + 255) >> 8) * ((shape->frame_height + 255) >> 8);
    /* With more than one group the LF groups are byte-aligned sections of their own: the LF coder is
     * then put in front of the entropy stage, so that its streams can be read back and wrapped
     * into sections on the host while the (2 ms, latency-bound) entropy stage is still running. */
    const int early_lf = hydamd_lf_coder(e->dev) && fg > 1;
    int ret = 0;
    if (early_lf)
        ret = hydamd_run_lf_coder(e->dev, (int)n, 1);
    if (!ret)
        ret = hydamd_finish_frame(e->dev, (int)n);
    if (!ret && staged_readback(e, n)) /* everything the host needs of this frame leaves the device as one blob */
        ret = hydamd_stage_frame_blob(e->dev, (int)n);
    return ret ? device_fail(e, ret) : 0;
}

/* ... and everything the frame assembler needs is read back and the frame written */
static int finish_frame_collect(HYDEncoder *e, const HydFrameShape *shape) {
    const size_t n = shape->lfg_count;
    const size_t fg = ((shape->frame_width + 255) >> 8) * ((shape->frame_height + 255) >> 8);
    const int lf_on_gpu = hydamd_lf_coder(e->dev);
    const int early_lf = lf_on_gpu && fg > 1;
    int staged = staged_readback(e, n); /* finish_frame_launch staged the frame's blob: one copy brings everything */
    const uint8_t *payload = NULL;      /* staged: the HF sections inside the blob's host copy */
    double t0 = now_ms();
    int ret = 0;
    LfgResult *res = calloc(n, sizeof(LfgResult));
    HydBits *lf_sections = NULL;
    HydAmdLfInfo *lf_info = NULL;
    uint8_t *lf_blob = NULL;
    size_t lf_len = 0;
    unsigned max_alphabet = 0;
    size_t payload_len = 0;
    if (!res)
        return FAIL(e, HYD_NOMEM, "out of memory");
    if (early_lf) {
        ret = hydamd_sync_lf(e->dev);
        TRACE("LF coder done", t0);
    } else {
        ret = hydamd_sync(e->dev);
        TRACE("GPU hot path (finish+sync)", t0);
    }
    if (ret) {
        ret = device_fail(e, ret);
        goto done;
    }
    t0 = now_ms();
    if (lf_on_gpu && (early_lf || !staged)) { /* two copies bring every LF group's coded coefficient stream */
        lf_len = hydamd_lf_payload_size(e->dev);
        lf_info = malloc(n * sizeof(HydAmdLfInfo));
        lf_blob = malloc(lf_len ? lf_len : 1);
        if (!lf_info || !lf_blob) {
            ret = FAIL(e, HYD_NOMEM, "out of memory");
            goto done;
        }
        ret = hydamd_read_lf_streams(e->dev, 0, (int)n, lf_info);
        if (!ret)
            ret = hydamd_read_lf_payload(e->dev, lf_blob, lf_len);
        for (size_t s = 0; s < n && !ret; s++) {
            if ((size_t)lf_info[s].offset + (((size_t)lf_info[s].bit_count + 7) >> 3) > lf_len) {
                ret = HYD_INTERNAL_ERROR;
                break;
            }
            memcpy(res[s].lf_lengths, lf_info[s].lengths, HYD_LF_CODES);
            res[s].lf_alphabet = lf_info[s].alphabet;
            res[s].lf_run_pairs = lf_info[s].run_pairs;
            res[s].lf_bit_count = lf_info[s].bit_count;
            res[s].lf_bits = lf_blob + lf_info[s].offset; /* borrowed from lf_blob */
        }
        if (ret) {
            ret = device_fail(e, ret);
            goto done;
        }
    }
    if (early_lf) {
        lf_sections = calloc(n, sizeof(HydBits));
        if (!lf_sections) {
            ret = FAIL(e, HYD_NOMEM, "out of memory");
            goto done;
        }
        ret = code_lf_groups_parallel(e, shape, res, lf_sections);
        TRACE("LF group sections (beside the entropy stage)", t0);
        if (ret)
            goto done;
        t0 = now_ms();
        ret = hydamd_sync(e->dev);
        TRACE("entropy stage done", t0);
        if (ret) {
            ret = device_fail(e, ret);
            goto done;
        }
        t0 = now_ms();
    }
    payload_len = hydamd_payload_size(e->dev);
    if (staged) {
        const void *blob = NULL;
        size_t bsize = 0;
        ret = hydamd_read_frame_blob(e->dev, (int)n, &blob, &bsize);
        const HydAmdBlobHeader *h = blob;
        if (!ret && (bsize < sizeof(*h) || h->magic != 0x42445948u || h->version != 1 || h->num_slots != n || h->lf_coded != 1 ||
                     h->total_bytes > bsize || (h->status & HYDAMD_BLOB_RETRY)))
            staged = 0; /* e.g. the frame outgrew a buffer and was rerun inside hydamd_sync: the staged blob is the first run's */
        if (!ret && staged) {
            const HydAmdBlobSlot *rec = (const HydAmdBlobSlot *)(h + 1);
            const uint64_t lf_off = sizeof(*h) + (uint64_t)n * sizeof(HydAmdBlobSlot);
            const int sane = lf_off <= h->total_bytes && h->lf_bytes <= h->total_bytes - lf_off && h->hf_bytes <= h->total_bytes &&
                             ((lf_off + h->lf_bytes + 15u) & ~(uint64_t)15u) + h->hf_bytes == h->total_bytes;
            if (!sane)
                ret = HYD_INTERNAL_ERROR;
            const uint8_t *lf_bytes = (const uint8_t *)blob + lf_off;
            for (size_t s = 0; s < n && !ret; s++) {
                if (rec[s].table_error || rec[s].lf.error ||
                    (uint64_t)rec[s].lf.offset + (((uint64_t)rec[s].lf.bit_count + 7) >> 3) > h->lf_bytes) {
                    ret = HYD_INTERNAL_ERROR;
                    break;
                }
                memcpy(res[s].freq, rec[s].freq, sizeof(res[s].freq));
                memcpy(res[s].alphabet, rec[s].alphabet, sizeof(res[s].alphabet));
                memcpy(res[s].bits, rec[s].group_bits, sizeof(res[s].bits));
                if (rec[s].running_max_alphabet > max_alphabet)
                    max_alphabet = rec[s].running_max_alphabet;
                if (!early_lf) {
                    memcpy(res[s].lf_lengths, rec[s].lf.lengths, HYD_LF_CODES);
                    res[s].lf_alphabet = rec[s].lf.alphabet;
                    res[s].lf_run_pairs = rec[s].lf.run_pairs;
                    res[s].lf_bit_count = rec[s].lf.bit_count;
                    res[s].lf_bits = (uint8_t *)(uintptr_t)(lf_bytes + rec[s].lf.offset); /* borrowed from the context's pinned copy */
                }
            }
            payload = (const uint8_t *)blob + h->total_bytes - h->hf_bytes;
            payload_len = (size_t)h->hf_bytes;
        }
    }
    if (!ret && !staged && lf_on_gpu && !early_lf && !lf_info) { /* (the staged blob was stale: the LF streams the old way) */
        lf_len = hydamd_lf_payload_size(e->dev);
        lf_info = malloc(n * sizeof(HydAmdLfInfo));
        lf_blob = malloc(lf_len ? lf_len : 1);
        if (!lf_info || !lf_blob) {
            ret = FAIL(e, HYD_NOMEM, "out of memory");
            goto done;
        }
        ret = hydamd_read_lf_streams(e->dev, 0, (int)n, lf_info);
        if (!ret)
            ret = hydamd_read_lf_payload(e->dev, lf_blob, lf_len);
        for (size_t s = 0; s < n && !ret; s++) {
            if ((size_t)lf_info[s].offset + (((size_t)lf_info[s].bit_count + 7) >> 3) > lf_len) {
                ret = HYD_INTERNAL_ERROR;
                break;
            }
            memcpy(res[s].lf_lengths, lf_info[s].lengths, HYD_LF_CODES);
            res[s].lf_alphabet = lf_info[s].alphabet;
            res[s].lf_run_pairs = lf_info[s].run_pairs;
            res[s].lf_bit_count = lf_info[s].bit_count;
            res[s].lf_bits = lf_blob + lf_info[s].offset;
        }
    }
    for (size_t s = 0; s < n && !ret && !staged; s++) {
        const size_t vbw = (shape->lfg[s].width + 7) >> 3, vbh = (shape->lfg[s].height + 7) >> 3;
        uint32_t log_alpha = 0, running = 0;
        if (!lf_on_gpu) {
            res[s].dc = malloc(3 * vbw * vbh * sizeof(int32_t));
            if (!res[s].dc) {
                ret = HYD_NOMEM;
                break;
            }
            ret = hydamd_read_dc(e->dev, (int)s, res[s].dc, vbw, vbh);
        }
        if (!ret)
            ret = hydamd_read_tables(e->dev, (int)s, res[s].freq, res[s].alphabet, &log_alpha, &running);
        if (!ret)
            ret = hydamd_read_sections(e->dev, (int)s, res[s].bits, NULL);
        if (running > max_alphabet)
            max_alphabet = running;
    }
    if (ret) {
        ret = device_fail(e, ret);
        goto done;
    }
    TRACE("read back results", t0);
    t0 = now_ms();
    ret = assemble_frame(e, shape, res, max_alphabet, staged ? payload : NULL, payload_len, lf_sections, NULL);
    TRACE("assemble frame (host)", t0);
done:
    for (size_t s = 0; s < n; s++) {
        free(res[s].dc);
        if (lf_sections)
            hb_free(&lf_sections[s]);
    }
    free(lf_sections);
    free(res);
    free(lf_info);
    free(lf_blob);
    return ret;
}

static int finish_frame(HYDEncoder *e, const HydFrameShape *shape) {
    const size_t fg = ((shape->frame_width + 255) >> 8) * ((shape->frame_height + 255) >> 8);
    if (shape->one_frame && fg > 1 && hydamd_lf_coder(e->dev) && !host_assembly_forced())
        return finish_frame_on_device(e, shape);
    const int ret = finish_frame_launch(e, shape);
    return ret ? ret : finish_frame_collect(e, shape);
}

/* ---- one frame on several devices (SURVEY 8(e) behind the C boundary) ----
 * hands the shards' contexts back; e->dev was only an alias of one of them */
/* A sharded frame's shards meet through PEER READS — the alphabet floor (hydamd_alphabet_floor_from_peers) and the blobs
 * the assembling device reads in place — whose visibility rests on a system-scope release event and the reading GPU's L2
 * invalidate at kernel start: reasoning until a node has run it.  So the path checks itself at first use,
 * the way the register LUT evaluation does: the FIRST sharded frame of the process over each (assembling entry, owning
 * entry) pair of the device list is verified — the floor against a host copy of the owners' maxima (hydamd_verify_floor),
 * every shard's view by a checksum on the owning and on the assembling device (hydamd_verify_enqueue) — and a pair that
 * passes is latched (cost once: ~4 ms for a 16384^2 frame).  A pair that FAILS: one line on stderr, that frame is finished
 * through host memory instead (finish_frame_through_host: same bytes, no peer read), and from then on the process keeps
 * every frame on one device.  HYDAMD_VERIFY_PEERS=1: every sharded frame is verified and a mismatch is an error that
 * names the device pair; =0: never. */
enum { VERIFY_NEVER = 0, VERIFY_ALWAYS = 1, VERIFY_FIRST_USE = 2 };
static int verify_peers_mode(void) {
    static int mode = -1;
    if (mode < 0) {
        const char *v = getenv("HYDAMD_VERIFY_PEERS");
        mode = !v || !*v ? VERIFY_FIRST_USE : *v == '0' ? VERIFY_NEVER : VERIFY_ALWAYS;
    }
    return mode;
}
static unsigned char g_pair_ok[HYD_MAX_DEVICES][HYD_MAX_DEVICES]; /* [assembling list entry][owning list entry], under g_ctx_lock */
static int g_peer_reads_bad;                                       /* a first-use verification failed: no more sharded frames */

static void multi_release(HYDEncoder *e) {
    if (e->shards > 1) {
        for (int d = 0; d < e->shards; d++) {
            struct Shard *sh = &e->multi[d];
            if (sh->dev)
                ctx_release(sh->dev, sh->slots, e->dev_linear, !sh->failed);
        }
        e->dev = NULL;
    }
    memset(e->multi, 0, sizeof(e->multi));
    e->shards = 0;
}

/* A sharded frame finished WITHOUT a peer read (the first-use verification of this frame's peer reads failed): every shard
 * gets its alphabet floor as a host value — the maximum over the earlier shards' per-LF-group maxima, each copied by the
 * host from its own device — and runs its stages again (hydamd_replay_frame); every shard's tables, section sizes, LF
 * streams and HF sections are read back from ITS device, and the frame is put together by the host's own writers
 * (assemble_frame), the sections in one piece per shard.  One GPU's rate at best; the reference's bytes. */
static int finish_frame_through_host(HYDEncoder *e, const HydFrameShape *shape, HydAmdContext *const *ctxs) {
    const int N = e->shards;
    const size_t n = shape->lfg_count;
    int ret = 0;
    unsigned max_alphabet = 0;
    uint32_t floor_so_far = 0;
    LfgResult *res = calloc(n, sizeof(LfgResult));
    HydAmdLfInfo *lf_info = malloc(n * sizeof(HydAmdLfInfo));
    uint8_t *lf_blob[HYD_MAX_DEVICES] = {0}, *hf[HYD_MAX_DEVICES] = {0};
    size_t hf_len[HYD_MAX_DEVICES] = {0};
    const uint8_t *seg_ptr[HYD_MAX_DEVICES];
    double t0 = now_ms();
    if (!res || !lf_info) {
        ret = FAIL(e, HYD_NOMEM, "out of memory");
        goto done;
    }
    for (int d = 0; d < N && !ret; d++) {
        HydAmdContext *c = ctxs[d];
        const int slots = (int)e->multi[d].slots;
        e->dev = c;
        if (d > 0) { /* the running maximum of entropy.c:459-460 over everything sent before this shard, as a host value */
            if ((ret = hydamd_set_alphabet_floor_device(c, NULL)) != 0 || (ret = hydamd_set_alphabet_floor(c, floor_so_far)) != 0 ||
                (ret = hydamd_replay_frame(c)) != 0)
                break;
        }
        if ((ret = hydamd_sync(c)) != 0)
            break;
        for (int i = 0; i < slots && !ret; i++) {
            uint32_t m = 0;
            ret = hydamd_read_alphabet_max(c, i, &m);
            floor_so_far = m > floor_so_far ? m : floor_so_far;
        }
        if (ret)
            break;
        LfgResult *r = res + e->multi[d].first_slot;
        HydAmdLfInfo *li = lf_info + e->multi[d].first_slot;
        const size_t lf_len = hydamd_lf_payload_size(c);
        lf_blob[d] = malloc(lf_len ? lf_len : 1);
        hf_len[d] = hydamd_payload_size(c);
        hf[d] = malloc(hf_len[d] ? hf_len[d] : 1);
        if (!lf_blob[d] || !hf[d]) {
            ret = FAIL(e, HYD_NOMEM, "out of memory");
            goto done;
        }
        if ((ret = hydamd_read_lf_streams(c, 0, slots, li)) != 0 || (ret = hydamd_read_lf_payload(c, lf_blob[d], lf_len)) != 0 ||
            (ret = hydamd_read_payload(c, hf[d], hf_len[d])) != 0)
            break;
        for (int i = 0; i < slots && !ret; i++) {
            uint32_t log_alpha = 0, running = 0;
            if ((size_t)li[i].offset + (((size_t)li[i].bit_count + 7) >> 3) > lf_len) {
                ret = HYD_INTERNAL_ERROR;
                break;
            }
            memcpy(r[i].lf_lengths, li[i].lengths, HYD_LF_CODES);
            r[i].lf_alphabet = li[i].alphabet;
            r[i].lf_run_pairs = li[i].run_pairs;
            r[i].lf_bit_count = li[i].bit_count;
            r[i].lf_bits = lf_blob[d] + li[i].offset; /* borrowed from lf_blob[d] */
            if ((ret = hydamd_read_tables(c, i, r[i].freq, r[i].alphabet, &log_alpha, &running)) == 0)
                ret = hydamd_read_sections(c, i, r[i].bits, NULL);
            if (running > max_alphabet)
                max_alphabet = running;
        }
        seg_ptr[d] = hf[d];
    }
    if (ret) {
        ret = ret == HYD_NOMEM ? ret : device_fail(e, ret);
        goto done;
    }
    TRACE("every shard again, floors and results through the host", t0);
    t0 = now_ms();
    {
        const PayloadSegments segs = {(size_t)N, seg_ptr, hf_len};
        size_t hf_total = 0;
        for (int d = 0; d < N; d++)
            hf_total += hf_len[d];
        ret = assemble_frame(e, shape, res, max_alphabet, NULL, hf_total, NULL, &segs);
    }
    TRACE("assemble frame (host)", t0);
done:
    for (int d = 0; d < N; d++) {
        free(lf_blob[d]);
        free(hf[d]);
    }
    free(lf_info);
    free(res);
    return ret;
}

/* The closing stage of a frame whose LF groups sit on several devices' contexts (shard d: tiles first_slot[d] ... in send
 * order).  Per device what the reference does per LF group (encoder.c:928-957), with two crossings, both device-side: the
 * running alphabet maximum (hydamd_alphabet_floor_from_peers: a peer read behind the earlier shards' transform kernels)
 * and the frame itself, which the first shard's GPU — the encoder's home device — assembles from every shard's blob, read
 * in place over xGMI (hydamd_wait_for + peer access); nothing of the frame passes through host memory before the
 * finished file. */
static int finish_frame_multi(HYDEncoder *e, const HydFrameShape *shape) {
    const int N = e->shards;
    const size_t n = shape->lfg_count;
    HydAmdContext *ctxs[HYD_MAX_DEVICES];
    double t0 = now_ms();
    int ret = 0;
    for (int d = 0; d < N; d++) {
        ctxs[d] = e->multi[d].dev;
        if (!ctxs[d])
            return FAIL(e, HYD_INTERNAL_ERROR, "a shard of this frame never received a tile");
    }
    /* which of this frame's peer reads are checked: all of them (HYDAMD_VERIFY_PEERS=1), or those of a (reader, owner)
     * pair of list entries no earlier frame of the process has verified */
    const int mode = verify_peers_mode();
    int check_view[HYD_MAX_DEVICES] = {0}, check_floor[HYD_MAX_DEVICES] = {0}, checking = 0;
    if (mode != VERIFY_NEVER) {
        pthread_mutex_lock(&g_ctx_lock);
        for (int d = 1; d < N; d++) {
            check_view[d] = mode == VERIFY_ALWAYS || !g_pair_ok[e->multi[0].entry][e->multi[d].entry];
            for (int p = 0; p < d; p++) /* shard d's floor kernel reads shards 0 .. d-1 */
                check_floor[d] |= mode == VERIFY_ALWAYS || !g_pair_ok[e->multi[d].entry][e->multi[p].entry];
            checking |= check_view[d] | check_floor[d];
        }
        pthread_mutex_unlock(&g_ctx_lock);
    }
    for (int d = 0; d < N; d++) { /* whatever of the transform stage the tiles' own calls did not enqueue */
        e->dev = ctxs[d];
        if ((ret = hydamd_run_transform(ctxs[d], (int)e->multi[d].slots)) != 0)
            return device_fail(e, ret);
    }
    for (int d = 1; d < N; d++) { /* shard d's entropy tables start from the maximum over shards 0 .. d-1 */
        e->dev = ctxs[d];
        if ((ret = hydamd_alphabet_floor_from_peers(ctxs[d], d, ctxs)) != 0)
            return device_fail(e, ret);
    }
    for (int d = 0; d < N; d++) {
        e->dev = ctxs[d];
        if ((ret = hydamd_finish_frame(ctxs[d], (int)e->multi[d].slots)) != 0)
            return device_fail(e, ret);
    }
    e->dev = ctxs[0];
    HydAmdAssembler *as = hydamd_context_assembler(ctxs[0]);
    if (!as)
        return device_fail(e, HYD_INTERNAL_ERROR);
    uint32_t lf_ids[HYDAMD_MAX_LF_GROUPS], blob_slots[HYD_MAX_DEVICES];
    for (size_t s = 0; s < n; s++)
        lf_ids[s] = (uint32_t)shape->lfg[s].raster_id;
    for (int d = 0; d < N; d++)
        blob_slots[d] = (uint32_t)e->multi[d].slots;
    ret = hydamd_assembler_plan(as, &e->metadata, 0, 1, (size_t)N, blob_slots, lf_ids, NULL, 0);
    if (ret)
        return FAIL(e, ret, "frame description rejected by the assembler");
    size_t out_cap = 4096 * n + (256u << 10);
    for (int d = 0; d < N; d++)
        out_cap += hydamd_blob_bound(ctxs[d], (int)e->multi[d].slots);
    unsigned reruns[HYD_MAX_DEVICES];
    for (int d = 0; d < N; d++)
        reruns[d] = hydamd_overflow_reruns(ctxs[d]);
    for (int attempt = 0; attempt < 4 + 2 * N; attempt++) {
        const void *blob[HYD_MAX_DEVICES];
        size_t cap[HYD_MAX_DEVICES], size = 0;
        for (int d = 0; d < N; d++) {
            e->dev = ctxs[d];
            if ((ret = hydamd_export_frame_owned(ctxs[d], (int)e->multi[d].slots, &blob[d], &cap[d])) != 0)
                return device_fail(e, ret);
        }
        e->dev = ctxs[0];
        for (int d = 1; d < N; d++) /* the assembling GPU's stream waits for the other shards' exports and may read their memory */
            if ((ret = hydamd_wait_for(ctxs[0], ctxs[d])) != 0)
                return device_fail(e, ret);
        for (int d = 1; d < N; d++) /* a shard's view summed where it was written and where it is about to be read */
            if (check_view[d]) {
                e->dev = ctxs[d];
                if ((ret = hydamd_verify_enqueue(ctxs[d], ctxs[d], (int)e->multi[d].slots, 0)) != 0)
                    return device_fail(e, ret);
                e->dev = ctxs[0];
                if ((ret = hydamd_verify_enqueue(ctxs[0], ctxs[d], (int)e->multi[d].slots, d)) != 0)
                    return device_fail(e, ret);
            }
        e->dev = ctxs[0];
        ret = hydamd_assembler_run(as, blob, cap, hydamd_get_stream(ctxs[0]), NULL, out_cap);
        if (ret)
            return device_fail(e, ret);
        for (int d = N - 1; d >= 0; d--) { /* a shard whose frame outgrew its buffers reruns it in here: its blob is then stale */
            e->dev = ctxs[d];
            if ((ret = hydamd_sync(ctxs[d])) != 0)
                return device_fail(e, ret);
        }
        /* a shard that reran its frame had left INCOMPLETE alphabet maxima the first time (a group that runs out of token
         * space stops counting): the later shards read their floor from those.  They read it again and run again. */
        int stale_from = 0;
        for (int d = 0; d < N; d++) {
            const unsigned now = hydamd_overflow_reruns(ctxs[d]);
            if (now != reruns[d] && !stale_from && d + 1 < N)
                stale_from = d + 1;
            reruns[d] = now;
        }
        if (stale_from) {
            for (int d = stale_from; d < N; d++) {
                e->dev = ctxs[d];
                if ((ret = hydamd_alphabet_floor_from_peers(ctxs[d], d, ctxs)) != 0 || (ret = hydamd_replay_frame(ctxs[d])) != 0)
                    return device_fail(e, ret);
            }
            continue;
        }
        TRACE("GPU hot path on every device + assembly", t0);
        t0 = now_ms();
        int asm_failed = 0;
        ret = hydamd_assembler_result(as, &size);
        if (ret) {
            const char *m = hydamd_assembler_error(as);
            if (m && strstr(m, "incomplete"))
                continue;
            if (ret == HYD_NEED_MORE_OUTPUT && size > out_cap) {
                out_cap = size;
                continue;
            }
            if (m && strstr(m, "NaN"))
                return FAIL(e, HYD_API_ERROR, "Invalid NaN Float");
            if (checking && mode == VERIFY_FIRST_USE) /* an assembler that read garbage through a bad peer mapping: let the checks decide */
                asm_failed = 1;
            else {
                mark_device_failed(e);
                return FAIL(e, ret < HYD_ERROR_START ? ret : HYD_INTERNAL_ERROR, "GPU frame assembly failed");
            }
        }
        int bad_reader = -1, bad_owner = -1, bad_shard = -1;
        const char *bad_what = "";
        for (int d = 1; d < N && bad_shard < 0; d++) {
            if (check_floor[d]) {
                int ok = 0;
                e->dev = ctxs[d];
                if ((ret = hydamd_verify_floor(ctxs[d], d, ctxs, &ok)) != 0)
                    return device_fail(e, ret);
                if (!ok) {
                    bad_reader = d, bad_owner = 0, bad_shard = d, bad_what = "alphabet floor";
                    break;
                }
            }
            if (check_view[d]) {
                unsigned long long written = 0, seen = 0;
                e->dev = ctxs[d];
                if ((ret = hydamd_verify_read(ctxs[d], 0, &written)) != 0)
                    return device_fail(e, ret);
                e->dev = ctxs[0];
                if ((ret = hydamd_verify_read(ctxs[0], d, &seen)) != 0)
                    return device_fail(e, ret);
                if (written != seen)
                    bad_reader = 0, bad_owner = d, bad_shard = d, bad_what = "frame view";
            }
        }
        e->dev = ctxs[0];
        if (bad_shard >= 0) {
            snprintf(e->verify_msg, sizeof(e->verify_msg), "peer read mismatch: device %d did not see what device %d wrote (shard %d, %s)",
                     hydamd_context_device(ctxs[bad_reader]), hydamd_context_device(ctxs[bad_owner]), bad_shard, bad_what);
            if (mode == VERIFY_ALWAYS) {
                mark_device_failed(e);
                return FAIL(e, HYD_INTERNAL_ERROR, e->verify_msg);
            }
            pthread_mutex_lock(&g_ctx_lock);
            const int first = !g_peer_reads_bad;
            g_peer_reads_bad = 1;
            pthread_mutex_unlock(&g_ctx_lock);
            if (first || trace_on())
                fprintf(stderr, "[hydrium] %s: this frame is finished through host memory, later frames stay on one device\n", e->verify_msg);
            return finish_frame_through_host(e, shape, ctxs);
        }
        if (checking) { /* every peer read of this frame was seen to return what its owner wrote: these pairs are trusted from here on */
            pthread_mutex_lock(&g_ctx_lock);
            for (int d = 1; d < N; d++) {
                if (check_view[d])
                    g_pair_ok[e->multi[0].entry][e->multi[d].entry] = 1;
                for (int p = 0; p < d && check_floor[d]; p++)
                    g_pair_ok[e->multi[d].entry][e->multi[p].entry] = 1;
            }
            pthread_mutex_unlock(&g_ctx_lock);
            TRACE("peer reads verified (first use of these device pairs)", t0);
            t0 = now_ms();
        }
        if (asm_failed) { /* the peer reads were fine: the assembly failed for a reason of its own */
            mark_device_failed(e);
            return FAIL(e, HYD_INTERNAL_ERROR, "GPU frame assembly failed");
        }
        uint8_t *dst = hb_extend(&e->stream, size);
        if (!dst)
            return FAIL(e, HYD_NOMEM, "out of memory");
        ret = hydamd_assembler_read(as, dst, size);
        TRACE("frame to the host", t0);
        return ret ? device_fail(e, ret) : 0;
    }
    mark_device_failed(e);
    return FAIL(e, HYD_INTERNAL_ERROR, "frame still does not fit after enlarging its buffers");
}

/* tile mode: the ring entry's frame, launched some calls ago, into the output stream */
static int pipe_collect(HYDEncoder *e, struct PendingTile *p) {
    struct PendingTile *const before = e->cur_pend;
    e->dev = p->dev;
    e->dev_failed = 0;
    e->cur_pend = p; /* a device error in here is this entry's */
    const int ret = finish_frame_collect(e, &p->shape);
    p->failed |= e->dev_failed;
    p->active = 0;
    e->cur_pend = before;
    return ret;
}

static void pipe_release(HYDEncoder *e) {
    for (int i = 0; i < TILE_PIPE_MAX; i++) {
        struct PendingTile *p = &e->pipe[i];
        if (!p->dev)
            continue;
        if (p->active && hydamd_sync(p->dev)) /* an abandoned image: its frames in flight are dropped */
            p->failed = 1;
        if (p->dev == e->dev && e->dev_failed)
            p->failed = 1;
        ctx_release(p->dev, e->dev_slots, e->dev_linear, !p->failed);
        memset(p, 0, sizeof(*p));
    }
    if (e->pipe_depth)
        e->dev = NULL; /* it was one of the ring's */
    e->pipe_depth = 0;
    e->cur_pend = NULL;
    e->tile_seq = 0;
}

/* additive API (include/hydrium_amd.h): tile frames in flight for this encoder, 1 (the reference's timing) .. 8;
 * 0 returns to the process default (HYDAMD_TILE_PIPELINE, else 1).  Not while frames are in flight. */
HYDRIUM_EXPORT int hydamd_set_tile_pipeline(HYDEncoder *e, int depth) {
    if (!e)
        return HYD_API_ERROR;
    if (depth < 0 || depth > TILE_PIPE_MAX)
        return FAIL(e, HYD_API_ERROR, "tile pipeline depth out of range");
    for (int i = 0; i < TILE_PIPE_MAX; i++)
        if (e->pipe[i].active)
            return FAIL(e, HYD_API_ERROR, "tile frames are in flight");
    if (e->have_metadata && e->one_frame) {
        /* the ring belongs to tile mode: a one-frame image has no use for it, and its context(s) — one per shard, tiles
         * already uploaded — are not the ring's to hand back */
        e->pipe_request = depth;
        return HYD_OK;
    }
    if (depth != e->pipe_request) {
        /* the ring is indexed by the depth: hand its contexts back, the next tile builds the new one */
        pipe_release(e);
        if (e->shards > 1) { /* (an encoder whose metadata changed from a sharded one-frame image) */
            multi_release(e);
        } else if (e->dev) { /* the encoder's own context (depth 1): the ring's entries bring theirs */
            ctx_release(e->dev, e->dev_slots, e->dev_linear, !e->dev_failed);
            e->dev = NULL;
        }
        e->dev_failed = 0;
    }
    e->pipe_request = depth;
    return HYD_OK;
}

HYDRIUM_EXPORT int hydamd_get_tile_pipeline(const HYDEncoder *e) { return e ? tile_pipeline_depth(e) : 0; }

HYDRIUM_EXPORT HYDStatusCode hyd_send_tile(HYDEncoder *e, const void *const buffer[3], uint32_t tile_x, uint32_t tile_y,
                                           ptrdiff_t row_stride, ptrdiff_t pixel_stride, int is_last,
                                           HYDSampleFormat sample_fmt) {
    int ret;
    if (sample_fmt != HYD_UINT8 && sample_fmt != HYD_UINT16 && sample_fmt != HYD_FLOAT32)
        return FAIL(e, HYD_API_ERROR, "Invalid Sample Format");
    if (!e->have_metadata)
        return FAIL(e, HYD_API_ERROR, "metadata was never set");
    /* geometry and last-tile rule: encoder.c:437-485 */
    const size_t W = e->metadata.width, H = e->metadata.height;
    if (tile_x >= (W + e->tile_w - 1) / e->tile_w || tile_y >= (H + e->tile_h - 1) / e->tile_h)
        return FAIL(e, HYD_API_ERROR, "tile out of bounds");
    if (e->one_frame && (e->frame_done || e->tiles_sent >= e->lfg_per_frame))
        return FAIL(e, HYD_API_ERROR, "the final tile of this image was already sent");
    if (e->one_frame && e->sent_mask[(size_t)tile_y * e->lfg_count_x + tile_x])
        return FAIL(e, HYD_API_ERROR, "this tile was already sent");
    const size_t tw = ((size_t)tile_x + 1) * e->tile_w > W ? W - tile_x * e->tile_w : e->tile_w;
    const size_t th = ((size_t)tile_y + 1) * e->tile_h > H ? H - tile_y * e->tile_h : e->tile_h;
    e->last_tile = is_last < 0 ? ((size_t)tile_x + 1) * e->tile_w >= W && ((size_t)tile_y + 1) * e->tile_h >= H : !!is_last;

    ret = emit_file_header(e);
    if (ret)
        return ret;

    device_list_init();
    if (!e->have_home) { /* encoders take the devices in turn */
        pthread_mutex_lock(&g_ctx_lock);
        e->home_index = (int)(g_encoder_seq++ % (unsigned long)g_device_count);
        e->home_device = g_devices[e->home_index];
        pthread_mutex_unlock(&g_ctx_lock);
        e->have_home = 1;
    }
    if (!e->shards) {
        const size_t n = e->lfg_per_frame;
        const int by_device = g_device_count, by_size = (int)(n / 2 < HYD_MAX_DEVICES ? n / 2 : HYD_MAX_DEVICES);
        pthread_mutex_lock(&g_ctx_lock);
        const int peers_bad = g_peer_reads_bad; /* a first-use verification of the peer reads failed earlier in this process */
        pthread_mutex_unlock(&g_ctx_lock);
        e->shards = e->one_frame && by_device > 1 && n >= (size_t)g_shard_min && !host_assembly_forced() && !peers_bad ? (by_device < by_size ? by_device : by_size) : 1;
        /* shard d runs on list entry (home + d) mod count: the first shard — whose GPU also assembles the frame and sends it
         * to the host — is the encoder's home device, which encoders take in turn, so concurrent sharded encoders do not all
         * assemble on the first GPU of the list */
        int devs[HYD_MAX_DEVICES];
        for (int d = 0; d < e->shards; d++)
            devs[d] = g_devices[(e->home_index + d) % g_device_count];
        if (e->shards > 1 && !hydamd_peers_reachable(devs, e->shards)) {
            /* a frame's shards meet through peer reads (finish_frame_multi): asked HERE, before the first tile is uploaded,
             * not when the last one arrives.  Without peer access between all of the devices the frame stays on this
             * encoder's home device (same bytes, one GPU's rate) */
            static int said;
            if (!said++ || trace_on())
                fprintf(stderr, "[hydrium] no peer access between the %d devices of this frame: coded on device %d alone\n",
                        e->shards, e->home_device);
            e->shards = 1;
        }
        memset(e->multi, 0, sizeof(e->multi));
        for (int d = 0; d < e->shards && e->shards > 1; d++) {
            e->multi[d].first_slot = (size_t)d * n / (size_t)e->shards;
            e->multi[d].slots = (size_t)(d + 1) * n / (size_t)e->shards - e->multi[d].first_slot;
            e->multi[d].entry = (e->home_index + d) % g_device_count;
        }
        if (e->shards > 1 && trace_on())
            fprintf(stderr, "[hydrium] frame dealt to %d shards, list entries %d.. (mod %d), assembled on entry %d = device %d\n",
                    e->shards, e->home_index, g_device_count, e->home_index, e->home_device);
    }
    if (e->shards > 1) { /* one frame dealt to several devices: this tile goes to the shard that owns its place in send order */
        const size_t slot = e->tiles_sent;
        int d = e->shards - 1;
        while (d > 0 && slot < e->multi[d].first_slot)
            d--;
        struct Shard *sh = &e->multi[d];
        e->cur_pend = NULL;
        e->dev_linear = e->metadata.linear_light != 0;
        if (!sh->dev) {
            int st = 0;
            const double tc = now_ms();
            sh->dev = ctx_acquire(g_devices[sh->entry], sh->slots, e->dev_linear, &st);
            TRACE("acquire device context (shard)", tc);
            if (!sh->dev) {
                const char *m = hydamd_error(NULL);
                if (st == HYD_NOMEM)
                    return FAIL(e, HYD_NOMEM, "out of device memory");
                return FAIL(e, HYD_INTERNAL_ERROR, m && strstr(m, "no usable HIP device")
                                                       ? "no usable HIP device (this build has no CPU fallback)"
                                                       : "GPU initialisation failed");
            }
            e->dev = sh->dev;
            if ((ret = hydamd_begin_frame(sh->dev, (unsigned)e->lfg_per_frame)) != 0)
                return device_fail(e, ret);
        }
        e->dev = sh->dev;
        const int local = (int)(slot - sh->first_slot);
        HydFrameLfg *l = &e->sent[slot];
        l->raster_id = (size_t)tile_y * e->lfg_count_x + tile_x;
        l->x = tile_x;
        l->y = tile_y;
        l->width = tw;
        l->height = th;
        const double tu = now_ms();
        ret = hydamd_encode_lf_group_host(sh->dev, local, buffer, row_stride, pixel_stride, (int)sample_fmt, tw, th, (unsigned)l->raster_id);
        if (!ret && eager_on()) {
            ret = hydamd_submit_lf_group(sh->dev, local);
            if (!ret && (local & 3) == 3 && !e->last_tile && hydamd_lf_coder(sh->dev))
                ret = hydamd_run_lf_coder(sh->dev, local + 1, 0);
        }
        if (ret)
            return device_fail(e, ret);
        TRACE("stage + upload tile (shard)", tu);
        e->sent_mask[l->raster_id] = 1;
        e->tiles_sent++;
        if (!e->last_tile) {
            drain(e);
            return HYD_OK;
        }
        if (e->tiles_sent != e->lfg_per_frame)
            return FAIL(e, HYD_API_ERROR, "one-frame mode needs every tile before the final one");
        HydFrameShape shape;
        memset(&shape, 0, sizeof(shape));
        shape.one_frame = 1;
        shape.image_width = shape.frame_width = W;
        shape.image_height = shape.frame_height = H;
        shape.tile_count_x = e->tile_w >> 8;
        shape.tile_count_y = e->tile_h >> 8;
        shape.lfg_count = e->lfg_per_frame;
        shape.lfg = e->sent;
        shape.is_last = e->last_tile;
        if ((ret = finish_frame_multi(e, &shape)) != 0)
            return ret;
        e->frame_done = 1;
        if (!e->out)
            return FAIL(e, HYD_API_ERROR, "buffer was never provided");
        drain(e);
        return e->stream_pos < e->stream.len ? HYD_NEED_MORE_OUTPUT : HYD_OK;
    }

    struct PendingTile *pend = NULL;
    e->cur_pend = NULL;
    if (!e->one_frame && tile_pipeline_depth(e) > 1) {
        /* this tile's ring entry: the frame it still holds is the oldest in flight */
        e->pipe_depth = tile_pipeline_depth(e);
        pend = &e->pipe[e->tile_seq % (size_t)e->pipe_depth];
        if (pend->active) {
            ret = pipe_collect(e, pend);
            if (ret)
                return ret;
        }
        e->dev = pend->dev;
        e->dev_failed = pend->failed;
        e->cur_pend = pend;
    }
    if (!e->dev) {
        int st = 0;
        const double tc = now_ms();
        e->dev_slots = e->lfg_per_frame;
        e->dev_linear = e->metadata.linear_light != 0;
        e->dev_failed = 0;
        e->dev = ctx_acquire(e->home_device, e->dev_slots, e->dev_linear, &st);
        TRACE("acquire device context", tc);
        if (!e->dev) {
            const char *m = hydamd_error(NULL);
            if (st == HYD_NOMEM)
                return FAIL(e, HYD_NOMEM, "out of device memory");
            return FAIL(e, HYD_INTERNAL_ERROR, m && strstr(m, "no usable HIP device")
                                                   ? "no usable HIP device (this build has no CPU fallback)"
                                                   : "GPU initialisation failed");
        }
    }
    if (pend)
        pend->dev = e->dev; /* the ring owns it from here on, whatever happens to this tile */
    const size_t slot = e->one_frame ? e->tiles_sent : 0;
    if (slot == 0) {
        const double tb = now_ms();
        ret = hydamd_begin_frame(e->dev, (unsigned)e->lfg_per_frame);
        if (ret)
            return device_fail(e, ret);
        TRACE("begin frame", tb);
    }
    HydFrameLfg *l = &e->sent[slot];
    l->raster_id = e->one_frame ? (size_t)tile_y * e->lfg_count_x + tile_x : 0;
    l->x = tile_x;
    l->y = tile_y;
    l->width = tw;
    l->height = th;
    const double tu = now_ms();
    ret = hydamd_encode_lf_group_host(e->dev, (int)slot, buffer, row_stride, pixel_stride, (int)sample_fmt, tw, th,
                                      (unsigned)l->raster_id);
    if (!ret && eager_on()) { /* work on this LF group while the caller prepares / we stage the next tile */
        ret = hydamd_submit_lf_group(e->dev, (int)slot);
        /* and every fourth tile the LF coder for the four just transformed: a launch of it takes
         * 0.25 ms whether it codes one LF group or sixteen, four tiles take 2 ms to stage.  Not behind the frame's final
         * tile: there nothing is left to hide it, and the closing stage runs it on a side stream beside the
         * (2 ms, latency-bound) entropy stage instead of in front of it */
        if (!ret && e->one_frame && (slot & 3) == 3 && !e->last_tile && hydamd_lf_coder(e->dev))
            ret = hydamd_run_lf_coder(e->dev, (int)slot + 1, 0);
    }
    if (ret)
        return device_fail(e, ret);
    TRACE("stage + upload tile", tu);

    if (e->one_frame) {
        e->sent_mask[l->raster_id] = 1;
        e->tiles_sent++;
        if (!e->last_tile) {
            drain(e);
            return HYD_OK; /* nothing of the frame exists yet; hyd_flush is a no-op here too (libhydrium.c:148-149) */
        }
        if (e->tiles_sent != e->lfg_per_frame)
            return FAIL(e, HYD_API_ERROR, "one-frame mode needs every tile before the final one");
    }

    HydFrameShape shape;
    memset(&shape, 0, sizeof(shape));
    shape.one_frame = e->one_frame;
    shape.image_width = W;
    shape.image_height = H;
    shape.frame_width = e->one_frame ? W : tw;
    shape.frame_height = e->one_frame ? H : th;
    shape.tile_count_x = e->tile_w >> 8;
    shape.tile_count_y = e->tile_h >> 8;
    shape.lfg_count = e->lfg_per_frame;
    shape.lfg = e->sent;
    shape.is_last = e->last_tile;
    if (pend) {
        pend->dev = e->dev;
        pend->failed = e->dev_failed;
        pend->lfg = e->sent[0];
        pend->shape = shape;
        pend->shape.lfg = &pend->lfg;
        e->tile_seq++;
        const double tl = now_ms();
        ret = finish_frame_launch(e, &pend->shape);
        TRACE("launch tile frame", tl);
        pend->failed |= e->dev_failed;
        if (ret)
            return ret;
        pend->active = 1;
        if (e->last_tile) { /* the image ends here: every frame in flight, oldest first */
            for (int i = 0; i < e->pipe_depth; i++) {
                struct PendingTile *q = &e->pipe[(e->tile_seq + (size_t)i) % (size_t)e->pipe_depth];
                if (q->active && (ret = pipe_collect(e, q)) != 0)
                    return ret;
            }
        }
    } else {
        ret = finish_frame(e, &shape);
        if (ret)
            return ret;
    }
    if (e->one_frame)
        e->frame_done = 1;
    /* the reference ends the tile with hyd_flush (encoder.c:1008): its error when no buffer is on loan
     * reaches the caller (libhydrium.c:193-195); HYD_NEED_MORE_OUTPUT is what the header documents */
    if (!e->out)
        return FAIL(e, HYD_API_ERROR, "buffer was never provided");
    drain(e);
    return e->stream_pos < e->stream.len ? HYD_NEED_MORE_OUTPUT : HYD_OK;
}

/* ---------------------------------------------------------------------------------------------
 * ICC pre-mangling (libhydrium.c:205-305): the JPEG XL ICC transform with an empty tag list and a
 * single "copy the rest" command, i.e. header prediction only.
 * ------------------------------------------------------------------------------------------- */

static uint8_t icc_header_guess(const uint8_t *icc, uint32_t icc_size, unsigned i) {
    static const char std12[] = "mntrRGB XYZ ";
    if (i < 4)
        return (uint8_t)(icc_size >> (8 * (3 - i)));
    if (i == 8)
        return 4;
    if (i >= 12 && i < 24)
        return (uint8_t)std12[i - 12];
    if (i >= 36 && i < 40)
        return (uint8_t)"acsp"[i - 36];
    if (i >= 41 && i < 44) {
        if (icc[40] == 'A')
            return (uint8_t)"PPL"[i - 41];
        if (icc[40] == 'M')
            return (uint8_t)"SFT"[i - 41];
        /* 'SGI ' / 'SUNW': bytes 42 and 43 are predicted from byte 41, which a decoder has only once it
         * has decoded position 41 — so position 41 itself takes the default prediction below */
        if (i >= 42 && icc[40] == 'S' && icc[41] == 'G')
            return (uint8_t)"I "[i - 42];
        if (i >= 42 && icc[40] == 'S' && icc[41] == 'U')
            return (uint8_t)"NW"[i - 42];
    }
    switch (i) {
    case 70:
        return 246;
    case 71:
        return 214;
    case 73:
        return 1;
    case 78:
        return 211;
    case 79:
        return 45;
    default:
        break;
    }
    if (i >= 80 && i < 84)
        return icc[i - 76];
    return 0;
}

HYDRIUM_EXPORT HYDStatusCode hyd_set_suggested_icc_profile(HYDEncoder *e, const uint8_t *icc_data, size_t icc_size) {
    if (!icc_data && !icc_size) {
        free(e->icc);
        e->icc = NULL;
        e->icc_size = 0;
        return HYD_OK;
    }
    if (!e->one_frame)
        return FAIL(e, HYD_API_ERROR, "one-frame mode required to set the suggested ICC profile");
    if (!icc_size || !icc_data || icc_size > UINT32_MAX)
        return FAIL(e, HYD_API_ERROR, "invalid ICC size or data buffer");
    HydBits b;
    hb_init(&b);
    const size_t head = icc_size < 128 ? icc_size : 128;
    const size_t rest = icc_size - head;
    hb_icc_varint(&b, icc_size);
    hb_icc_varint(&b, rest ? 3 + (size_t)(63 - __builtin_clzll(rest)) / 7 : 0);
    if (rest) {
        hb_icc_varint(&b, 0); /* empty tag list */
        hb_put(&b, 1, 8);     /* command 1: copy */
        hb_icc_varint(&b, rest);
    }
    hb_align(&b);
    for (unsigned i = 0; i < head; i++)
        hb_put(&b, (uint8_t)(icc_data[i] - icc_header_guess(icc_data, (uint32_t)icc_size, i)), 8);
    hb_align(&b);
    hb_append_bytes(&b, icc_data + head, rest);
    if (b.failed) {
        hb_free(&b);
        return FAIL(e, HYD_NOMEM, "out of memory");
    }
    free(e->icc);
    e->icc = b.data; /* ownership moves to the encoder */
    e->icc_size = b.len;
    return HYD_OK;
}

/* ---------------------------------------------------------------------------------------------
 * Additive entry point (include/hydrium_amd.h): wrap LF-group results that were produced elsewhere
 * — on other GPUs of the node, or by several contexts — into one frame.  This is the same
 * assemble_frame() hyd_send_tile ends in; it touches no GPU.
 * ------------------------------------------------------------------------------------------- */
static int frame_from_parts(const HYDImageMetadata *md, int write_header, int is_last, size_t lfg_count,
                            const uint32_t *tile_xy, const int32_t *const *dc, const HydAmdLfStream *lf, const uint32_t *freq,
                            const uint32_t *alphabet, const uint32_t *group_bits, unsigned max_alphabet,
                            const uint8_t *payload, size_t payload_len, const PayloadSegments *segs, const uint8_t *icc,
                            size_t icc_size, uint8_t **out, size_t *out_len, const char **err) {
    HYDEncoder *e = hyd_encoder_new();
    if (!e)
        return HYD_NOMEM;
    int ret = hyd_set_metadata(e, md);
    if (!ret && icc)
        ret = hyd_set_suggested_icc_profile(e, icc, icc_size);
    if (!ret && lfg_count != e->lfg_per_frame)
        ret = FAIL(e, HYD_API_ERROR, "a frame needs every one of its LF groups");
    LfgResult *res = calloc(lfg_count ? lfg_count : 1, sizeof(LfgResult));
    if (!ret && !res)
        ret = HYD_NOMEM;
    if (!ret)
        take_spare_buffer_for(&e->stream, payload_len + (payload_len >> 3) + 65536); /* also when no header is written */
    if (!ret && write_header)
        ret = emit_file_header(e);
    if (!ret) {
        const size_t W = md->width, H = md->height;
        for (size_t s = 0; s < lfg_count; s++) {
            const size_t tx = tile_xy[2 * s], ty = tile_xy[2 * s + 1];
            if (tx >= (W + e->tile_w - 1) / e->tile_w || ty >= (H + e->tile_h - 1) / e->tile_h) {
                ret = FAIL(e, HYD_API_ERROR, "tile out of bounds");
                break;
            }
            e->sent[s].raster_id = e->one_frame ? ty * e->lfg_count_x + tx : 0;
            if (e->sent_mask[e->sent[s].raster_id]) {
                ret = FAIL(e, HYD_API_ERROR, "an LF group appears twice in the frame description");
                break;
            }
            e->sent_mask[e->sent[s].raster_id] = 1;
            e->sent[s].x = tx;
            e->sent[s].y = ty;
            e->sent[s].width = (tx + 1) * e->tile_w > W ? W - tx * e->tile_w : e->tile_w;
            e->sent[s].height = (ty + 1) * e->tile_h > H ? H - ty * e->tile_h : e->tile_h;
            if (lf) { /* LF coefficients already coded (on a GPU): borrowed, like dc */
                if (!lf[s].lengths || (!lf[s].bits && lf[s].bit_count) || lf[s].bit_count > UINT32_MAX) {
                    ret = FAIL(e, HYD_API_ERROR, "incomplete LF stream");
                    break;
                }
                memcpy(res[s].lf_lengths, lf[s].lengths, HYD_LF_CODES);
                res[s].lf_bits = (uint8_t *)(lf[s].bits ? lf[s].bits : (const uint8_t *)"");
                res[s].lf_alphabet = lf[s].alphabet;
                res[s].lf_run_pairs = lf[s].run_pairs;
                res[s].lf_bit_count = (uint32_t)lf[s].bit_count;
            } else {
                res[s].dc = (int32_t *)dc[s];
            }
            memcpy(res[s].freq, freq + s * HYD_FRAME_MAX_CLUSTERS * HYD_FRAME_ALPHABET, sizeof(res[s].freq));
            memcpy(res[s].alphabet, alphabet + s * HYD_FRAME_MAX_CLUSTERS, sizeof(res[s].alphabet));
            memcpy(res[s].bits, group_bits + s * HYDAMD_GROUPS_PER_LFG, sizeof(res[s].bits));
        }
    }
    if (!ret) {
        HydFrameShape shape;
        memset(&shape, 0, sizeof(shape));
        shape.one_frame = e->one_frame;
        shape.image_width = md->width;
        shape.image_height = md->height;
        shape.frame_width = e->one_frame ? md->width : e->sent[0].width;
        shape.frame_height = e->one_frame ? md->height : e->sent[0].height;
        shape.tile_count_x = e->tile_w >> 8;
        shape.tile_count_y = e->tile_h >> 8;
        shape.lfg_count = lfg_count;
        shape.lfg = e->sent;
        shape.is_last = is_last;
        ret = assemble_frame(e, &shape, res, max_alphabet, payload, payload_len, NULL, segs);
    }
    if (!ret) {
        hb_align(&e->stream);
        if (e->stream.data && !e->stream.failed) { /* the caller takes the stream's buffer itself (hydamd_free = free) */
            *out = e->stream.data;
            *out_len = e->stream.len;
            e->stream.data = NULL;
            e->stream.len = e->stream.cap = 0;
        } else {
            ret = HYD_NOMEM;
        }
    }
    if (err)
        *err = e->error;
    free(res);
    hyd_encoder_destroy(e);
    return ret;
}

HYDRIUM_EXPORT int hydamd_frame_from_results(const HYDImageMetadata *md, int write_header, int is_last, size_t lfg_count,
                                             const uint32_t *tile_xy, const int32_t *const *dc, const uint32_t *freq,
                                             const uint32_t *alphabet, const uint32_t *group_bits, unsigned max_alphabet,
                                             const uint8_t *payload, size_t payload_len, const uint8_t *icc,
                                             size_t icc_size, uint8_t **out, size_t *out_len, const char **err) {
    if (!dc)
        return HYD_API_ERROR;
    return frame_from_parts(md, write_header, is_last, lfg_count, tile_xy, dc, NULL, freq, alphabet, group_bits, max_alphabet,
                            payload, payload_len, NULL, icc, icc_size, out, out_len, err);
}

HYDRIUM_EXPORT int hydamd_frame_from_streams(const HYDImageMetadata *md, int write_header, int is_last, size_t lfg_count,
                                             const uint32_t *tile_xy, const HydAmdLfStream *lf, const uint32_t *freq,
                                             const uint32_t *alphabet, const uint32_t *group_bits, unsigned max_alphabet,
                                             const uint8_t *payload, size_t payload_len, const uint8_t *icc,
                                             size_t icc_size, uint8_t **out, size_t *out_len, const char **err) {
    if (!lf)
        return HYD_API_ERROR;
    return frame_from_parts(md, write_header, is_last, lfg_count, tile_xy, NULL, lf, freq, alphabet, group_bits, max_alphabet,
                            payload, payload_len, NULL, icc, icc_size, out, out_len, err);
}

/* From the blobs hydamd_export_frame leaves on the GPUs that coded the frame's LF groups (copied to
 * host memory by the caller): the same assembly, no GPU. */
HYDRIUM_EXPORT int hydamd_frame_from_blobs(const HYDImageMetadata *md, int write_header, int is_last, size_t nblobs,
                                           const void *const *blobs, const size_t *blob_sizes, const uint8_t *icc,
                                           size_t icc_size, uint8_t **out, size_t *out_len, const char **err) {
    static const char *const bad = "malformed LF-group blob";
    if (!md || !blobs || !blob_sizes || !out || !out_len || !nblobs) {
        if (err)
            *err = "null argument";
        return HYD_API_ERROR;
    }
    if (!md->width || !md->height) { /* before the LF-group grid below divides by its width */
        if (err)
            *err = "invalid zero-width or zero-height";
        return HYD_API_ERROR;
    }
    size_t slots = 0, hf_total = 0;
    for (size_t b = 0; b < nblobs; b++) {
        const HydAmdBlobHeader *h = blobs[b];
        if (!h || blob_sizes[b] < sizeof(*h) || h->magic != 0x42445948u || h->version != 1 || h->total_bytes > blob_sizes[b] ||
            (h->status & HYDAMD_BLOB_RETRY) || h->lf_coded != 1) { /* 0: LF ints not coded; 0x101: a view, for device assemblers only */
            if (err)
                *err = h && blob_sizes[b] >= sizeof(*h) && (h->status & HYDAMD_BLOB_RETRY)
                           ? "a blob is incomplete (its frame outgrew a buffer): rerun that shard"
                           : bad;
            return HYD_API_ERROR;
        }
        const uint64_t lf_off = sizeof(*h) + (uint64_t)h->num_slots * sizeof(HydAmdBlobSlot);
        /* every size is checked against the blob's own length before it enters a sum: a damaged lf_bytes near
         * 2^64 must not wrap hf_off back into range */
        const int sane = lf_off <= h->total_bytes && h->lf_bytes <= h->total_bytes - lf_off && h->hf_bytes <= h->total_bytes;
        const uint64_t hf_off = sane ? (lf_off + h->lf_bytes + 15u) & ~(uint64_t)15u : 0;
        if (!sane || hf_off > h->total_bytes || hf_off + h->hf_bytes != h->total_bytes) {
            if (err)
                *err = bad;
            return HYD_API_ERROR;
        }
        if (h->status & 1u) {
            if (err)
                *err = "Invalid NaN Float";
            return HYD_API_ERROR;
        }
        slots += h->num_slots;
        hf_total += (size_t)h->hf_bytes;
    }
    const size_t lfg_x = (md->width + 2047) >> 11;
    uint32_t *tile_xy = malloc((slots ? slots : 1) * 2 * sizeof(uint32_t));
    HydAmdLfStream *lf = calloc(slots ? slots : 1, sizeof(HydAmdLfStream));
    uint32_t *freq = malloc((slots ? slots : 1) * sizeof(((HydAmdBlobSlot *)0)->freq));
    uint32_t *alpha = malloc((slots ? slots : 1) * sizeof(((HydAmdBlobSlot *)0)->alphabet));
    uint32_t *bits = malloc((slots ? slots : 1) * sizeof(((HydAmdBlobSlot *)0)->group_bits));
    /* a frame of one group splices its only section bit by bit and wants it in one piece; any other
     * frame takes the sections piece by piece, straight from the blobs */
    const int one_group = ((md->width + 255) >> 8) * ((md->height + 255) >> 8) == 1;
    uint8_t *payload = one_group ? malloc(hf_total ? hf_total : 1) : NULL;
    const uint8_t **seg_ptr = malloc(nblobs * sizeof(*seg_ptr));
    size_t *seg_len = malloc(nblobs * sizeof(*seg_len));
    int ret = HYD_OK;
    unsigned max_alphabet = 0;
    if (!tile_xy || !lf || !freq || !alpha || !bits || (one_group && !payload) || !seg_ptr || !seg_len) {
        ret = HYD_NOMEM;
        if (err)
            *err = "out of memory";
    }
    size_t s = 0, hf_pos = 0;
    for (size_t b = 0; b < nblobs && !ret; b++) {
        const HydAmdBlobHeader *h = blobs[b];
        const HydAmdBlobSlot *rec = (const HydAmdBlobSlot *)(h + 1);
        const uint8_t *lf_bytes = (const uint8_t *)blobs[b] + sizeof(*h) + (size_t)h->num_slots * sizeof(HydAmdBlobSlot);
        const uint8_t *hf_bytes = (const uint8_t *)blobs[b] + h->total_bytes - h->hf_bytes;
        for (uint32_t i = 0; i < h->num_slots && !ret; i++, s++) {
            if (rec[i].table_error || rec[i].lf.error ||
                (uint64_t)rec[i].lf.offset + (((uint64_t)rec[i].lf.bit_count + 7) >> 3) > h->lf_bytes) {
                ret = HYD_INTERNAL_ERROR;
                if (err)
                    *err = rec[i].table_error ? "ANS table construction failed on the device"
                                              : rec[i].lf.error ? "LF code construction failed on the device" : bad;
                break;
            }
            tile_xy[2 * s] = (uint32_t)(rec[i].preset % lfg_x);
            tile_xy[2 * s + 1] = (uint32_t)(rec[i].preset / lfg_x);
            lf[s].lengths = rec[i].lf.lengths;
            lf[s].alphabet = rec[i].lf.alphabet;
            lf[s].run_pairs = rec[i].lf.run_pairs;
            lf[s].bits = lf_bytes + rec[i].lf.offset;
            lf[s].bit_count = rec[i].lf.bit_count;
            memcpy(freq + s * HYD_FRAME_MAX_CLUSTERS * HYD_FRAME_ALPHABET, rec[i].freq, sizeof(rec[i].freq));
            memcpy(alpha + s * HYD_FRAME_MAX_CLUSTERS, rec[i].alphabet, sizeof(rec[i].alphabet));
            memcpy(bits + s * HYDAMD_GROUPS_PER_LFG, rec[i].group_bits, sizeof(rec[i].group_bits));
            if (rec[i].running_max_alphabet > max_alphabet)
                max_alphabet = rec[i].running_max_alphabet;
        }
        if (payload)
            memcpy(payload + hf_pos, hf_bytes, (size_t)h->hf_bytes);
        seg_ptr[b] = hf_bytes;
        seg_len[b] = (size_t)h->hf_bytes;
        hf_pos += (size_t)h->hf_bytes;
    }
    if (!ret) {
        const PayloadSegments segs = {nblobs, seg_ptr, seg_len};
        ret = frame_from_parts(md, write_header, is_last, slots, tile_xy, NULL, lf, freq, alpha, bits, max_alphabet, payload,
                               hf_total, one_group ? NULL : &segs, icc, icc_size, out, out_len, err);
    }
    free((void *)seg_ptr);
    free(seg_len);
    free(tile_xy);
    free(lf);
    free(freq);
    free(alpha);
    free(bits);
    free(payload);
    return ret;
}

/* ---- internal entry points of the device-side assembler's planner (assembler.c); not exported ---- */

/* the file header hyd_send_tile writes in front of this image's first frame */
int hyd_internal_file_header(const HYDImageMetadata *md, const uint8_t *icc, size_t icc_size, HydBits *out, const char **err) {
    HYDEncoder *e = hyd_encoder_new();
    if (!e)
        return HYD_NOMEM;
    int ret = hyd_set_metadata(e, md);
    if (!ret && icc)
        ret = hyd_set_suggested_icc_profile(e, icc, icc_size);
    if (!ret)
        ret = hyd_write_file_header(out, e->metadata.width, e->metadata.height, e->level10, e->icc, e->icc_size, &e->error);
    if (err)
        *err = e->error;
    hyd_encoder_destroy(e);
    return ret;
}

/* the geometry-only bits that close an LF group section (cached per shape for the life of the process) */
const HydBits *hyd_internal_lf_tail(size_t vbw, size_t vbh) { return lf_tail(vbw, vbh); }

HYDRIUM_EXPORT void hydamd_free(void *p) {
    if (!p)
        return;
    /* every buffer that leaves through *out is a malloc block: the allocator knows how much of it is usable */
    if (!offer_spare_buffer(p, malloc_usable_size(p)))
        free(p);
}

/* Release every device context parked by destroyed encoders (their device memory, pinned staging and
 * streams).  Encoders alive at the time keep theirs. */
HYDRIUM_EXPORT void hydamd_trim_cache(void) {
    HydAmdContext *victims[CTX_POOL_MAX];
    int n = 0;
    pthread_mutex_lock(&g_ctx_lock);
    for (int i = 0; i < CTX_POOL_MAX; i++)
        if (g_pool[i].ctx) {
            victims[n++] = g_pool[i].ctx;
            g_pool[i].ctx = NULL;
        }
    pthread_mutex_unlock(&g_ctx_lock);
    for (int i = 0; i < n; i++)
        hydamd_destroy(victims[i]);
    void *spares[SPARE_SLOTS];
    pthread_mutex_lock(&g_buf_lock);
    for (int i = 0; i < SPARE_SLOTS; i++) {
        spares[i] = g_spare[i].p;
        g_spare[i].p = NULL;
        g_spare[i].cap = 0;
    }
    pthread_mutex_unlock(&g_buf_lock);
    for (int i = 0; i < SPARE_SLOTS; i++)
        free(spares[i]);
}

/* the CPU-only tests drive the same function through libhydrium_hosttest.so */
#ifdef HYD_TEST_HOOKS
#define HYDT_EXPORT __attribute__((visibility("default")))
HYDT_EXPORT int hydt_frame_from_stages(const HYDImageMetadata *md, int write_header, int is_last, size_t lfg_count,
                                       const uint32_t *tile_xy, const int32_t *const *dc, const uint32_t *freq,
                                       const uint32_t *alphabet, const uint32_t *group_bits, unsigned max_alphabet,
                                       const uint8_t *payload, size_t payload_len, const uint8_t *icc, size_t icc_size,
                                       uint8_t **out, size_t *out_len, const char **err) {
    return hydamd_frame_from_results(md, write_header, is_last, lfg_count, tile_xy, dc, freq, alphabet, group_bits,
                                     max_alphabet, payload, payload_len, icc, icc_size, out, out_len, err);
}
HYDT_EXPORT void hydt_free(void *p) { hydamd_free(p); }

/* depth-limited code lengths of the host prefix coder (prefix.c), for comparison with the device's */
HYDT_EXPORT int hydt_code_lengths(const uint32_t *freq, uint32_t *lengths, uint32_t n, int max_depth) {
    return hps_code_lengths(freq, lengths, n, max_depth);
}

static int hydt_take(HydBits *b, int ret, uint8_t **out, size_t *out_len) {
    hb_align(b);
    if (!ret && b->failed)
        ret = HYD_NOMEM;
    if (!ret) {
        *out = malloc(b->len ? b->len : 1);
        if (*out) {
            memcpy(*out, b->data, b->len);
            *out_len = b->len;
        } else {
            ret = HYD_NOMEM;
        }
    }
    hb_free(b);
    return ret;
}

/* the pre-mangled ICC profile hyd_set_suggested_icc_profile keeps (what the file header then entropy-codes) */
HYDT_EXPORT int hydt_icc_mangled(const uint8_t *icc, size_t icc_size, uint8_t **out, size_t *out_len) {
    HYDEncoder *e = hyd_encoder_new();
    if (!e)
        return HYD_NOMEM;
    e->one_frame = 1;
    int ret = hyd_set_suggested_icc_profile(e, icc, icc_size);
    if (!ret) {
        *out = malloc(e->icc_size ? e->icc_size : 1);
        if (*out) {
            memcpy(*out, e->icc, e->icc_size);
            *out_len = e->icc_size;
        } else {
            ret = HYD_NOMEM;
        }
    }
    hyd_encoder_destroy(e);
    return ret;
}

/* one byte-padded LFGroup section from LF ints (host coder) ... */
HYDT_EXPORT int hydt_lf_group(const int32_t *dc, size_t vbw, size_t vbh, uint8_t **out, size_t *out_len) {
    HydBits b;
    const char *err = NULL;
    hb_init(&b);
    return hydt_take(&b, hyd_write_lf_group(&b, dc, vbw, vbh, &err), out, out_len);
}

/* ... and from an LF-coefficient stream coded elsewhere (the GPU, or the tests' model of it) */
HYDT_EXPORT int hydt_lf_group_coded(size_t vbw, size_t vbh, const uint8_t *lengths, uint32_t alphabet, uint32_t run_pairs,
                                    const uint8_t *bits, uint64_t bit_count, uint8_t **out, size_t *out_len) {
    HydBits b;
    const char *err = NULL;
    const HydLfCoded lf = {lengths, alphabet, run_pairs, bits, bit_count};
    hb_init(&b);
    return hydt_take(&b, hyd_write_lf_group_coded(&b, vbw, vbh, &lf, lf_tail(vbw, vbh), &err), out, out_len);
}
#endif /* HYD_TEST_HOOKS */
