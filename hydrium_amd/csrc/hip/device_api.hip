/*
 * device_api.hip — the thin C-ABI between the C host side and the HIP kernels (include/hydrium_amd.h).
 *
 * Memory plan.  Token storage is sized for the typical case and grown on demand, never on the hot
 * path of a frame that fits:
 *   per LF-group slot   tokens   64 groups x tok_cap records x 4 B (8 B once a float LF group is seen);
 *                                tok_cap = 98304 = 1.5 symbols per pixel             =  25 MB
 *                       aux      the lane-per-group entropy form's refill words, 2 B / record + 1 bit  =  13 MB
 *                       payload  packed HF sections, 1 byte per pixel                =   4 MB
 *                       tables   HydkTables                                          = 235 KB
 *                       dc       3 x 256 x 256 int32                                 = 768 KB
 *                       LF coder: records 196608 x 8 B + bits                        =   4.5 MB  (hard worst case)
 *   per context         in_lut8 (512 B), in_lut16 (128 KB), bias_lut (256 KB)
 *                       bitbuf   only if the wave-per-group entropy form is used: reversed bit buffers,
 *                                tok_cap x 46 bits per group                         =  36 MB per slot
 *                       2 x (pinned host + device) staging tiles for the host-pointer path
 * A 16384x16384 frame (64 slots) holds 3.0 GB, an 8192x8192 frame 0.75 GB.  A frame that needs more
 * (a group above 1.5 symbols per pixel: noise; or more than a byte per pixel of sections) raises a
 * status bit on the device; hydamd_sync() then enlarges the arrays to their hard maximum and runs
 * the frame again from the job descriptors it still holds (resolve_overflow).
 */
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "../../../include/hydrium_amd.h"
#include "hydk_common.h"

#pragma clang fp contract(off)

namespace hydk {
hipError_t launch_transform(const HydkLfJob *d_jobs, int num_slots, unsigned fmt_mask, int xmode, uint32_t *status,
                            uint2 *part_info, int plog, hipStream_t stream);
hipError_t launch_tables(const uint32_t *hist, HydkTables *tabs, const uint32_t *alpha_max, int nclusters, int first_slot,
                         int num_slots, uint32_t alpha_floor, const uint32_t *alpha_floor_dev, const uint32_t *lf_hist,
                         HydkLfStream *lf_streams, void *lf_work, int slots_per_frame, hipStream_t stream);
hipError_t launch_export(const HydkLfJob *d_jobs, const HydkTables *tabs, const uint32_t *group_bits, const HydkLfStream *lf_streams,
                         const uint8_t *payload, const uint64_t *hf_total, const uint8_t *lf_packed,
                         const unsigned long long *lf_total, const uint32_t *status, int num_slots, int lf_coded, uint8_t *dst,
                         uint64_t capacity, int view, hipStream_t stream);
hipError_t launch_rans(const HydkLfJob *d_jobs, const uint32_t *sym_count, const HydkTables *tabs, uint32_t *bitbuf,
                       uint32_t bit_pitch_words, uint32_t *group_bits, int preset_bits, int num_slots, const uint32_t *status,
                       hipStream_t stream);
hipError_t launch_rans_deferred(const HydkLfJob *d_jobs, const uint32_t *sym_count, const HydkTables *tabs, uint16_t *aux,
                                uint16_t *flags, uint32_t aux_pitch, uint32_t *final_state, uint32_t *group_bits,
                                int preset_bits, int num_slots, const uint32_t *status, hipStream_t stream);
hipError_t launch_rans_lanes(const HydkLfJob *d_jobs, const uint32_t *sym_count, const HydkTables *tabs, uint16_t *aux,
                             uint16_t *flags, uint32_t aux_pitch, uint32_t *final_state, uint32_t *group_bits, int preset_bits,
                             int nclusters, int num_slots, const uint32_t *status, const uint32_t *lf_hist,
                             HydkLfStream *lf_streams, void *lf_work, hipStream_t stream);
hipError_t launch_lf_front(const HydkLfJob *d_jobs, unsigned long long *recs, uint32_t *hist, void *work, int num_slots,
                           hipStream_t stream);
hipError_t launch_lf_back(const HydkLfJob *d_jobs, const unsigned long long *recs, HydkLfStream *streams, uint32_t *bits,
                          void *work, int num_slots, hipStream_t stream);
hipError_t launch_rans_emit(const HydkLfJob *d_jobs, const uint32_t *sym_count, const uint16_t *aux, const uint16_t *flags,
                            uint32_t aux_pitch, const uint32_t *final_state, const uint32_t *group_bits, const uint64_t *offsets,
                            uint8_t *payload, int preset_bits, int num_slots, const uint32_t *status, hipStream_t stream);
hipError_t launch_frame_begin(const HydkLfJob *host_jobs, HydkLfJob *d_jobs, int count, uint32_t *accum, size_t accum_words,
                              hipStream_t stream);
hipError_t launch_publish(const uint64_t *total, uint64_t *h_total, const unsigned long long *lf_total,
                          unsigned long long *h_lf_total, const uint32_t *status, uint32_t *h_status, hipStream_t stream);
hipError_t transform_footprint(int fmt, int xmode, int *lds_bytes, int *registers);
hipError_t launch_scan(const uint32_t *group_bits, int count, uint64_t *offsets, uint64_t *total, uint8_t *payload,
                       uint64_t payload_cap, int clear_shared_words, uint32_t *status, hipStream_t stream);
hipError_t launch_pack(const uint32_t *bitbuf, uint32_t bit_pitch_words, const uint32_t *group_bits, const uint64_t *offsets,
                       uint8_t *payload, int count, const uint32_t *status, hipStream_t stream);
size_t lf_work_bytes();
hipError_t launch_lf_coder(const HydkLfJob *d_jobs, unsigned long long *recs, uint32_t *hist, HydkLfStream *streams,
                           uint32_t *bits, void *work, int num_slots, hipStream_t stream);
hipError_t launch_lf_gather(HydkLfStream *streams, const uint32_t *bits, uint32_t *packed, unsigned long long *total,
                            int num_slots, hipStream_t stream);
hipError_t launch_lf_huffman_only(const uint32_t *hist, HydkLfStream *stream_out, uint32_t *codes, hipStream_t stream);
hipError_t launch_lut_selftest(const uint16_t *in_lut16, const float *bias_lut, int linear_light, int xmode,
                               uint32_t *mismatches, hipStream_t stream);
} // namespace hydk

/* HYDStatusCode values (include/libhydrium/libhydrium.h) */
#define ST_OK 0
#define ST_NOMEM (-13)
#define ST_API_ERROR (-14)
#define ST_INTERNAL_ERROR (-15)

static_assert(sizeof(HydAmdBlobHeader) == 64 && sizeof(HydAmdBlobSlot) == 16 + 36 + 12 + 256 + 4608 + sizeof(HydkLfStream) &&
                  sizeof(HydAmdBlobSlot) % 16 == 0,
              "blob records are laid out by k_export_frame word by word");
static_assert(HYDAMD_MAX_CLUSTERS == HYDK_MAX_CLUSTERS && HYDAMD_ALPHABET == HYDK_ALPHABET &&
                  HYDAMD_GROUPS_PER_LFG == HYDK_GROUPS_PER_LFG && HYDAMD_LF_CODES == HYDK_LF_CODES,
              "public and kernel-side table shapes must agree");

namespace {

constexpr int kStaging = 2;
constexpr int kSplitSlots = 2; /* transform launches of at most this many LF groups split every group over four workgroups */
/* HYDAMD_K1_SPLIT_SLOTS=n (A/B knob, bytes unchanged): launches of up to n LF groups are split instead — n = 32 makes the
 * pipelined loop's transform workgroups last a quarter as long (round 6: does a chain workgroup that waits for 80 KB of one
 * compute unit's LDS start sooner when the transform workgroups around it retire four times as often?) */
int split_slots() {
    static const int n = [] {
        const char *e = getenv("HYDAMD_K1_SPLIT_SLOTS");
        const int v = e ? atoi(e) : kSplitSlots;
        return v < 0 ? 0 : v > 256 ? 256 : v;
    }();
    return n;
}
constexpr size_t kDbgPlane = (size_t)2048 * 2048;

char g_global_error[256] = "";

struct TimedLaunch {
    int cls;
    hipEvent_t start, stop;
};

} // namespace

struct HydAmdContext {
    int device = 0;
    int max_slots = 0;
    int linear_light = 0;
    int use_luts = 2;               /* XYB mode: 0 registers + fast reciprocal, 1 registers + IEEE division, 2 LUT gathers */
    int best_register_mode = 2;     /* best mode that passed the bit-exactness self-test */
    int curve_gathers = 0;          /* hydamd_set_curve_gathers: 0 by the last frame's density, 1 always (round 4), 2 never */
    uint64_t published_pixels = 0;  /* pixels of the frame whose section total k_publish wrote last */
    uint64_t seen_bytes = 0, seen_pixels = 0; /* HF section bytes and pixels of the last frame hydamd_sync waited for: ONE frame's pair (curve-gather choice) */
    int register_luts_ok = 0;
    int rans_lanes = 0;             /* entropy-stage form: 0 wave per group (form 4), non-zero lane per group (form 5, also what 6 asks for; float frames still take form 4) */
    uint32_t tok_cap = HYDK_DEFAULT_TOKEN_CAP; /* token records per group the arrays below hold */
    bool caps_forced = false;       /* HYDAMD_TOKEN_CAP / HYDAMD_PAYLOAD_CAP sized the arrays (tests): no growing ahead */
    unsigned grown_ahead = 0;       /* times the arrays were enlarged before a frame because another context had outgrown the defaults */
    uint32_t rec_bytes = 4;         /* their record size: 4 until a float LF group is recorded, then 8 */
    uint32_t bit_pitch_words = 0;   /* words per group in bitbuf (0: not allocated yet) */
    int want_transform = 0, want_entropy = 0; /* what the caller asked for this frame: replayed if a buffer was too small */
    bool slot_lanes[HYDAMD_MAX_LF_GROUPS] = {}; /* which entropy form coded each slot of the current frame */
    unsigned overflow_reruns = 0;   /* frames run twice because a buffer was too small (hydamd_overflow_reruns) */
    uint32_t alpha_floor = 0;       /* running maximum alphabet of the LF groups coded before this context's */
    const uint32_t *alpha_floor_dev = nullptr; /* the same, left in device memory by the caller's exchange (or NULL) */
    unsigned num_presets = 1;
    int scheme = 0;
    int nclusters = 9;
    int preset_bits = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    char error[256] = "";

    /* device memory */
    char *tokens = nullptr;         /* [slots][64][tok_cap] records of rec_bytes */
    uint32_t *bitbuf = nullptr;     /* [slots][64][bit_pitch_words], wave form only, allocated on first use */
    uint16_t *rans_aux = nullptr;   /* [slots][64][tok_cap] lane form: the 16 bits a refill at each symbol sends */
    uint16_t *rans_flags = nullptr; /* [slots][64][tok_cap / 16] lane form: one refill flag per symbol */
    uint32_t *rans_final = nullptr; /* [slots][64] lane form: final states */
    uint32_t *rbits_total = nullptr; /* [slots][64] residue bits per group */
    HydkTables *tables = nullptr;   /* [slots] */
    int32_t *dc = nullptr;          /* [slots][3][256][256] */
    uint32_t *hist = nullptr;       /* [slots][9][128] */
    uint32_t *sym_count = nullptr;  /* [slots][64] */
    uint2 *part_info = nullptr;     /* [kSplitSlots][64][4] {symbols, residue bits} of a group's parts (transform launches of one or two LF groups) */
    uint32_t *group_bits = nullptr; /* [slots][64] */
    uint64_t *offsets = nullptr;    /* [slots][64] */
    uint64_t *total = nullptr;      /* [1] */
    uint32_t *status = nullptr;     /* [1] bit 0: non-finite float sample */
    uint32_t *alpha_max = nullptr;  /* [slots] largest token + 1 per LF group */
    HydkLfJob *d_jobs = nullptr;    /* [slots] */
    HydkLfJob *h_jobs = nullptr;    /* [slots] pinned mirror of the frame being submitted (one of the ring below) */
    HydkLfJob *h_jobs_ring[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t jobs_uploaded[4] = {nullptr, nullptr, nullptr, nullptr};
    unsigned frame_counter = 0;
    int jobs_idx = 0;
    uint16_t *in_lut8 = nullptr, *in_lut16 = nullptr;
    float *bias_lut = nullptr;
    uint8_t *payload = nullptr;
    size_t payload_cap = 0;

    /* LF-group coder (lf_coder.hip): runs on its own stream between two events of the main one */
    int lf_on_device = 1;                  /* 0 off, 1 on a side stream beside the entropy stage, 2 at the end of the main stream */
    unsigned long long *lf_recs = nullptr; /* [slots][HYDK_LF_SYMBOLS] */
    uint32_t *lf_hist = nullptr;           /* [slots + 1][HYDK_LF_CODES]; the last entry is hydamd_debug_lf_code's scratch */
    uint32_t *lf_codes = nullptr;          /* [HYDK_LF_CODES] likewise */
    char *lf_work = nullptr;               /* [slots] per-LF-group scratch of the LF coder's kernels (codes, window bit counts) */
    HydkLfStream *lf_streams = nullptr;    /* [slots + 1] */
    uint32_t *lf_bits = nullptr;           /* [slots][HYDK_LF_BITWORDS] */
    uint32_t *lf_packed = nullptr;         /* the same symbol data, back to back in slot order (4-byte aligned) */
    unsigned long long *lf_total = nullptr; /* [1] bytes in lf_packed */
    unsigned long long *h_lf_total_pinned = nullptr;
    uint64_t h_lf_total = 0;
    hipStream_t lf_stream = nullptr;
    hipEvent_t lf_fork = nullptr, lf_join = nullptr;
    bool lf_pending = false;               /* the side stream holds work the main stream has not waited for */
    bool lf_need_gather = false;           /* LF groups were coded since the frame's LF streams were last packed */
    bool lf_results_valid = false;         /* hydamd_sync_lf (or hydamd_sync) has seen the packed LF streams complete */
    hipEvent_t lf_ready = nullptr;         /* recorded behind the LF gather kernel and the copy of its byte count */
    int lf_slots = 0;                      /* slots covered by the last LF coder run */
    int transformed = 0, coded = 0, lf_coded = 0; /* slots of the current frame whose transform / entropy / LF kernels are enqueued */
    int host_staged = 0;                   /* slots [0, host_staged) of the current frame have pixels in the staging arena */
    float *dbg_xyb = nullptr, *dbg_dct = nullptr;
    int32_t *dbg_quant = nullptr;

    /* staging for the host-pointer path: pinned bounce tiles + one device tile per slot */
    void *pinned[kStaging] = {nullptr, nullptr};
    char *d_arena = nullptr;
    size_t arena_tile = 0;
    size_t staging_cap = 0;
    hipEvent_t staged[kStaging] = {nullptr, nullptr};
    int staging_next = 0;
    /* uploads run on their own stream so that tile n + 1 is copied while tile n's transform kernel runs */
    hipStream_t copy_stream = nullptr;
    bool copy_stream_shared = false; /* one of the device's shared upload streams (acquire_copy_stream): not this context's to destroy */
    hipEvent_t frame_fence = nullptr; /* recorded on the main stream at hydamd_begin_frame */
    bool copy_needs_fence = false;

    /* host mirrors filled by hydamd_sync */
    uint64_t h_total = 0;
    uint32_t h_status = 0;
    int slots_finished = 0;
    int slots_per_frame = 0; /* > 0: the slots hold a batch of independent frames of this many LF groups each (hydamd_begin_batch) */
    bool results_valid = false;
    uint32_t *accum = nullptr;       /* [hist | alpha_max | status | lf_hist]: cleared once per frame by k_frame_begin */
    size_t accum_words = 0;
    bool accum_stale = true;
    bool lf_total_unpublished = false;
    bool status_published = false;   /* nothing that can set status bits was enqueued after the last launch_publish */
    uint64_t *h_total_pinned = nullptr;
    uint32_t *h_status_pinned = nullptr;

    /* several contexts (devices) working on one frame: hydamd_wait_for, hydamd_alphabet_floor_from_peers */
    hipEvent_t peer_event = nullptr; /* "everything enqueued so far on this context's stream" */
    uint32_t *peer_floor = nullptr;  /* [1] the floor k_floor_from_peers leaves for the table kernel */
    unsigned long long *verify_sums = nullptr; /* [HYDAMD_MAX_PEERS] checksums of frame views (hydamd_verify_enqueue) */

    /* the drop-in API's frame assembly on the device (hydamd_export_frame_owned, hydamd_context_assembler) */
    void *own_blob = nullptr;
    size_t own_blob_cap = 0;
    void *stage_blob = nullptr;      /* hydamd_stage_frame_blob: the frame's self-contained blob in device memory ... */
    size_t stage_blob_cap = 0;
    uint8_t *stage_host = nullptr;   /* ... and where hydamd_read_frame_blob lands it (pinned) */
    size_t stage_host_cap = 0;
    int stage_slots = 0;
    HydAmdAssembler *assembler = nullptr;

    /* profiling */
    bool profiling = false;
    std::vector<TimedLaunch> timed;
    double prof_ms[HYDAMD_K_COUNT] = {0, 0, 0, 0, 0};
    uint64_t prof_n[HYDAMD_K_COUNT] = {0, 0, 0, 0, 0};
};

namespace {

int fail(HydAmdContext *ctx, int code, const char *what, hipError_t e = hipSuccess) {
    char *dst = ctx ? ctx->error : g_global_error;
    if (e != hipSuccess)
        snprintf(dst, 256, "%s: %s", what, hipGetErrorString(e));
    else
        snprintf(dst, 256, "%s", what);
    return code;
}

#define HIP_TRY(ctx, call)                                                          \
    do {                                                                            \
        hipError_t e__ = (call);                                                    \
        if (e__ != hipSuccess)                                                      \
            return fail(ctx, e__ == hipErrorOutOfMemory ? ST_NOMEM : ST_INTERNAL_ERROR, #call, e__); \
    } while (0)

/* ---- the three LUTs of the integer pixel path, built on the host exactly as the reference does
 * (format.c:15-36,58-83); x86-64 without contraction is the canonical arithmetic ---- */
float host_linearize(float x) {
    if (x <= 0.0404482362771082f)
        return 0.07739938080495357f * x;
    return 0.003094300919832f + x * (-0.009982599f + x * (0.72007737769f + 0.2852804880f * x));
}

float host_bias(float v) {
    const float x = v + 0.0037930732552754493f;
    union {
        float f;
        uint32_t u;
    } z;
    z.f = x;
    z.u = 0x548c39cbu - z.u / 3u;
    z.f *= 1.5015480449f - 0.534850249f * x * z.f * z.f * z.f;
    z.f *= 1.333333985f - 0.33333333f * x * z.f * z.f * z.f;
    return 1.0f / z.f - 0.155954f;
}

void host_input_lut(uint16_t *lut, size_t size, int linear_light) {
    const float unit = 1.0f / (size - 1.0f);
    for (size_t i = 0; i < size; i++) {
        const float f = i * unit;
        const int32_t y = (int32_t)((linear_light ? f : host_linearize(f)) * 65535.f + 0.5f);
        lut[i] = (uint16_t)(y < 0 ? 0 : y > 65535 ? 65535 : y);
    }
}

size_t sample_size(int fmt) { return fmt == HYDK_FMT_U8 ? 1 : fmt == HYDK_FMT_U16 ? 2 : 4; }

/* forms 1-3 (several chains per wavefront, by rows) were replaced by form 5: select it, and say so once */
int retired_rans_form(int waves) {
    static bool told = false;
    if (!told) {
        told = true;
        fprintf(stderr, "[hydrium] entropy-stage form %d no longer exists: using form 5 (one lane per group), its successor\n", waves);
    }
    return 1;
}

struct ScopedTimer {
    HydAmdContext *ctx;
    TimedLaunch tl;
    bool on;
    hipStream_t stream;
    ScopedTimer(HydAmdContext *c, int cls, hipStream_t s = nullptr) : ctx(c), on(c->profiling), stream(s ? s : c->stream) {
        if (!on)
            return;
        tl.cls = cls;
        (void)hipEventCreate(&tl.start);
        (void)hipEventCreate(&tl.stop);
        (void)hipEventRecord(tl.start, stream);
    }
    ~ScopedTimer() {
        if (!on)
            return;
        (void)hipEventRecord(tl.stop, stream);
        ctx->timed.push_back(tl);
    }
};

void drain_timers(HydAmdContext *ctx) {
    for (TimedLaunch &tl : ctx->timed) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, tl.start, tl.stop) == hipSuccess) {
            ctx->prof_ms[tl.cls] += ms;
            ctx->prof_n[tl.cls]++;
        }
        (void)hipEventDestroy(tl.start);
        (void)hipEventDestroy(tl.stop);
    }
    ctx->timed.clear();
}

int check_slot(HydAmdContext *ctx, int slot) {
    if (!ctx)
        return fail(nullptr, ST_API_ERROR, "null context");
    if (slot < 0 || slot >= ctx->max_slots)
        return fail(ctx, ST_API_ERROR, "LF-group slot out of range");
    return ST_OK;
}

/* where slot's token records live in the arrays as they are sized right now */
void bind_slot_buffers(HydAmdContext *ctx, int slot) {
    HydkLfJob &job = ctx->h_jobs[slot];
    job.tokens = ctx->tokens + (size_t)slot * HYDK_GROUPS_PER_LFG * ctx->tok_cap * ctx->rec_bytes;
    job.tok_cap = ctx->tok_cap;
    job.rec_bytes = ctx->rec_bytes;
}

int widen_token_records(HydAmdContext *ctx);

/* Record one LF group's job; the kernels run batched over all recorded slots in hydamd_finish_frame. */
int record_lf_group(HydAmdContext *ctx, int slot, const void *const src[3], ptrdiff_t row_stride, ptrdiff_t pixel_stride,
                    int fmt, size_t width, size_t height, unsigned preset) {
    if (width == 0 || height == 0 || width > 2048 || height > 2048)
        return fail(ctx, ST_API_ERROR, "LF group must be between 1 and 2048 pixels in each direction");
    if (fmt != HYDK_FMT_U8 && fmt != HYDK_FMT_U16 && fmt != HYDK_FMT_F32)
        return fail(ctx, ST_API_ERROR, "Invalid Sample Format");
    if (preset >= ctx->num_presets)
        return fail(ctx, ST_API_ERROR, "preset out of range for this frame");
    if (fmt == HYDK_FMT_F32 && ctx->rec_bytes != 8) { /* float tokens need the 8-byte record: once per context */
        const int st = widen_token_records(ctx);
        if (st != ST_OK)
            return st;
    }

    HydkLfJob &job = ctx->h_jobs[slot];
    memset(&job, 0, sizeof(job));
    for (int c = 0; c < 3; c++)
        job.src[c] = src[c];
    job.row_stride = row_stride;
    job.pixel_stride = pixel_stride;
    job.fmt = fmt;
    job.linear_light = ctx->linear_light;
    job.width = (int)width;
    job.height = (int)height;
    job.gcols = (int)((width + 255) >> 8);
    job.grows = (int)((height + 255) >> 8);
    job.scheme = ctx->scheme;
    job.use_luts = fmt == HYDK_FMT_F32 ? 0 : ctx->use_luts;
    job.preset = preset;
    job.in_lut8 = ctx->in_lut8;
    job.in_lut16 = ctx->in_lut16;
    job.bias_lut = ctx->bias_lut;
    bind_slot_buffers(ctx, slot);
    job.sym_count = ctx->sym_count + (size_t)slot * HYDK_GROUPS_PER_LFG;
    job.rbits_total = ctx->rbits_total + (size_t)slot * HYDK_GROUPS_PER_LFG;
    job.hist = ctx->hist + (size_t)slot * HYDK_MAX_CLUSTERS * HYDK_ALPHABET;
    job.alpha_max = ctx->alpha_max + slot;
    job.dc = ctx->dc + (size_t)slot * 3 * HYDK_DC_PITCH * HYDK_DC_PITCH;
    if (slot == 0) { /* the dump planes hold one LF group */
        job.dbg_xyb = ctx->dbg_xyb;
        job.dbg_dct = ctx->dbg_dct;
        job.dbg_quant = ctx->dbg_quant;
    }
    ctx->results_valid = false;
    return ST_OK;
}

/* Upload streams.  The uploads of all contexts end up on the same few DMA engines, and every stream a process creates
 * takes a turn in the runtime's rotation over the hardware queues (more active streams than about 22 queues and the
 * firmware time-slices: a batch of encoder threads, each with a kernel stream, an upload stream and an LF side stream,
 * fell from 1400 to 500 frames/s at 24).  So the contexts of a device share HYDAMD_COPY_STREAMS upload streams
 * (default 4, dealt in turn; 0 = one per context, as until round 4).  Ordering is by the contexts' own events. */
constexpr int kMaxSharedCopyStreams = 8;
static std::mutex g_copy_lock;
static hipStream_t g_copy_streams[HYDAMD_MAX_PEERS * 2][kMaxSharedCopyStreams];
static unsigned g_copy_next[HYDAMD_MAX_PEERS * 2];

int acquire_copy_stream(HydAmdContext *ctx) {
    static const int shared = [] {
        const char *v = getenv("HYDAMD_COPY_STREAMS");
        const int n = v && *v ? atoi(v) : 4;
        return n < 0 ? 0 : n > kMaxSharedCopyStreams ? kMaxSharedCopyStreams : n;
    }();
    if (!shared || ctx->device < 0 || ctx->device >= HYDAMD_MAX_PEERS * 2) {
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        return ST_OK;
    }
    std::lock_guard<std::mutex> hold(g_copy_lock);
    const unsigned k = g_copy_next[ctx->device]++ % (unsigned)shared;
    if (!g_copy_streams[ctx->device][k])
        HIP_TRY(ctx, hipStreamCreateWithFlags(&g_copy_streams[ctx->device][k], hipStreamNonBlocking));
    ctx->copy_stream = g_copy_streams[ctx->device][k];
    ctx->copy_stream_shared = true;
    return ST_OK;
}

int ensure_staging(HydAmdContext *ctx, size_t tile_bytes) {
    if (tile_bytes <= ctx->staging_cap)
        return ST_OK;
    /* A later tile of the frame has a wider sample type than the first (the reference lets sample_fmt
     * vary per tile): the arena grows.  Tiles already uploaded keep their pixels — they are still needed
     * if their transform kernels have not run yet (HYDAMD_EAGER=0) or have to run again (a frame that
     * outgrows its token arrays, or whose token records are widened for a float tile). */
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (!ctx->copy_stream) { /* taken on first use: device-pointer users never need it */
        const int st = acquire_copy_stream(ctx);
        if (st != ST_OK)
            return st;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->copy_stream));
    for (int i = 0; i < kStaging; i++) {
        if (ctx->pinned[i])
            (void)hipHostFree(ctx->pinned[i]);
        ctx->pinned[i] = nullptr;
    }
    for (int i = 0; i < kStaging; i++) {
        HIP_TRY(ctx, hipHostMalloc(&ctx->pinned[i], tile_bytes, hipHostMallocDefault));
        if (!ctx->staged[i])
            HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->staged[i], hipEventDisableTiming));
    }
    char *fresh = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&fresh, tile_bytes * (size_t)ctx->max_slots));
    if (ctx->d_arena) {
        for (int i = 0; i < ctx->host_staged; i++) {
            HydkLfJob &job = ctx->h_jobs[i];
            const char *old_base = ctx->d_arena + (size_t)i * ctx->arena_tile;
            if (!job.width || (const char *)job.src[0] != old_base)
                continue; /* not a host-staged slot */
            const size_t ss = sample_size(job.fmt);
            char *base = fresh + (size_t)i * tile_bytes;
            HIP_TRY(ctx, hipMemcpy(base, old_base, (size_t)job.width * job.height * 3 * ss, hipMemcpyDeviceToDevice));
            job.src[0] = base;
            job.src[1] = base + ss;
            job.src[2] = base + 2 * ss;
        }
        (void)hipFree(ctx->d_arena);
    }
    ctx->d_arena = fresh;
    ctx->arena_tile = tile_bytes;
    ctx->staging_cap = tile_bytes;
    return ST_OK;
}

/* interleave the caller's (possibly planar / strided / bottom-up) samples as packed RGB */
template <typename T>
void gather_rows(T *dst, const void *const src[3], ptrdiff_t row_stride, ptrdiff_t pixel_stride, size_t w, size_t y0,
                 size_t y1) {
    const T *r = (const T *)src[0], *g = (const T *)src[1], *b = (const T *)src[2];
    const bool interleaved = pixel_stride == 3 && g == r + 1 && b == r + 2;
    for (size_t y = y0; y < y1; y++) {
        T *d = dst + y * w * 3;
        const ptrdiff_t yo = (ptrdiff_t)y * row_stride;
        if (interleaved) {
            memcpy(d, r + yo, w * 3 * sizeof(T));
        } else {
            for (size_t x = 0; x < w; x++) {
                const ptrdiff_t o = yo + (ptrdiff_t)x * pixel_stride;
                d[3 * x] = r[o];
                d[3 * x + 1] = g[o];
                d[3 * x + 2] = b[o];
            }
        }
    }
}

/* A 2048 x 2048 RGB16 tile is 25 MB: one core copies it in about 1.2 ms, which made staging the
 * largest part of hyd_send_tile.  A few threads share the rows (HYDAMD_STAGE_THREADS, default up to 8). */
int stage_threads() {
    static int n = 0;
    if (!n) {
        const char *env = getenv("HYDAMD_STAGE_THREADS");
        n = env ? atoi(env) : 0;
        if (n < 1) {
            const unsigned hw = std::thread::hardware_concurrency();
            n = hw > 8 ? 8 : hw > 0 ? (int)hw : 1;
        }
        if (n > 32)
            n = 32;
    }
    return n;
}

/* Persistent helpers for the staging gather.  A tile takes a quarter of a millisecond to copy once
 * the rows are spread over a few cores — less than it costs to start the threads each time — so the
 * helpers are started once and parked on a condition variable.  Several gathers may be under way (encoders
 * on several threads): every gather is a job of row bands, claimed one at a time by its own caller and by
 * whichever helpers are free, so eight encoder threads share the helpers instead of seven of them copying
 * their 12 MB tiles alone (a batch of 4K frames on 8 threads: the slowest tenth of the tiles staged in 0.57 ms
 * instead of 0.96; the mean, 0.58, is the copy engine's queue and did not move). */
class StagePool {
  public:
    /* fn(band) for band = 0 .. bands - 1, on the calling thread and on up to `helpers` pool threads */
    template <typename Fn>
    void run(size_t bands, int helpers, Fn &&fn) {
        Job job;
        job.fn = [&fn](size_t band) { fn(band); };
        job.bands = bands;
        {
            std::lock_guard<std::mutex> g(m_);
            ensure_workers(helpers);
            jobs_.push_back(&job);
        }
        cv_work_.notify_all();
        work_on(job);
        std::unique_lock<std::mutex> lk(m_);
        for (size_t i = 0; i < jobs_.size(); i++) /* nothing left to claim: helpers stop looking at it */
            if (jobs_[i] == &job) {
                jobs_.erase(jobs_.begin() + (long)i);
                break;
            }
        cv_done_.wait(lk, [&] { return job.done.load() == job.bands; }); /* bands still in other hands */
    }

  private:
    struct Job {
        std::function<void(size_t)> fn;
        size_t bands = 0;
        std::atomic<size_t> next{0}, done{0};
    };
    /* The job sits on its caller's stack and is gone once `done` reaches `bands`: a thread touches it only while it
     * holds a band that is not yet counted — so the next band is claimed BEFORE the finished one is counted. */
    void work_from(Job *job, size_t b, const size_t bands) {
        for (;;) {
            job->fn(b);
            const size_t next = job->next.fetch_add(1);
            const bool last = job->done.fetch_add(1) + 1 == bands;
            if (last) {
                std::lock_guard<std::mutex> g(m_); /* the waiter checks under this lock */
                cv_done_.notify_all();
            }
            if (next >= bands)
                return;
            b = next;
        }
    }
    void work_on(Job &job) {
        const size_t b = job.next.fetch_add(1);
        if (b < job.bands)
            work_from(&job, b, job.bands);
    }
    void ensure_workers(int want) { /* m_ held */
        while ((int)workers_.size() < want) {
            try {
                workers_.emplace_back([this] { loop(); });
                workers_.back().detach(); /* parked for the life of the process */
            } catch (...) {
                break;
            }
        }
    }
    void loop() {
        std::unique_lock<std::mutex> lk(m_);
        for (;;) {
            Job *job = nullptr;
            cv_work_.wait(lk, [&] {
                for (Job *j : jobs_)
                    if (j->next.load() < j->bands) {
                        job = j;
                        return true;
                    }
                return false;
            });
            /* claim the first band under the lock: the job cannot be retired (its caller erases it under the same
             * lock, and then waits for `done`) between the look and the claim */
            const size_t bands = job->bands;
            const size_t b = job->next.fetch_add(1);
            lk.unlock();
            if (b < bands)
                work_from(job, b, bands);
            lk.lock();
        }
    }
    std::mutex m_;
    std::condition_variable cv_work_, cv_done_;
    std::vector<std::thread> workers_;
    std::vector<Job *> jobs_;
};

StagePool &stage_pool() {
    static StagePool *pool = new StagePool(); /* never destroyed: its detached helpers outlive main() */
    return *pool;
}

template <typename T>
void gather_packed(T *dst, const void *const src[3], ptrdiff_t row_stride, ptrdiff_t pixel_stride, size_t w, size_t h) {
    const size_t threads = (size_t)stage_threads();
    constexpr size_t kBand = 64; /* rows per claim */
    const size_t bands = (h + kBand - 1) / kBand;
    if (threads > 1 && bands > 1) {
        stage_pool().run(bands, (int)threads - 1, [&](size_t b) {
            gather_rows(dst, src, row_stride, pixel_stride, w, b * kBand, b * kBand + kBand < h ? b * kBand + kBand : h);
        });
        return;
    }
    gather_rows(dst, src, row_stride, pixel_stride, w, 0, h);
}

} // namespace

extern "C" {

/* memcpy for the host side's large moves (a finished 12 MB frame into the caller's buffer: 0.4 ms on one core), shared
 * with the staging helpers */
void hydamd_host_copy(void *dst, const void *src, size_t n) {
    constexpr size_t kPiece = (size_t)1 << 20;
    const size_t threads = (size_t)stage_threads();
    if (n < 4 * kPiece || threads < 2) {
        memcpy(dst, src, n);
        return;
    }
    const size_t pieces = (n + kPiece - 1) / kPiece;
    stage_pool().run(pieces, (int)threads - 1, [&](size_t b) {
        const size_t at = b * kPiece;
        memcpy((char *)dst + at, (const char *)src + at, n - at < kPiece ? n - at : kPiece);
    });
}

int hydamd_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

const char *hydamd_error(HydAmdContext *ctx) { return ctx ? ctx->error : g_global_error; }

void hydamd_destroy(HydAmdContext *ctx) {
    if (!ctx)
        return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream)
        (void)hipStreamSynchronize(ctx->stream);
    if (ctx->assembler)
        hydamd_assembler_destroy(ctx->assembler);
    if (ctx->own_blob)
        (void)hipFree(ctx->own_blob);
    if (ctx->stage_blob)
        (void)hipFree(ctx->stage_blob);
    if (ctx->stage_host)
        (void)hipHostFree(ctx->stage_host);
    drain_timers(ctx);
    if (ctx->copy_stream) {
        (void)hipStreamSynchronize(ctx->copy_stream);
        if (!ctx->copy_stream_shared)
            (void)hipStreamDestroy(ctx->copy_stream);
    }
    if (ctx->frame_fence)
        (void)hipEventDestroy(ctx->frame_fence);
    if (ctx->lf_stream) {
        (void)hipStreamSynchronize(ctx->lf_stream);
        (void)hipStreamDestroy(ctx->lf_stream);
    }
    if (ctx->lf_fork)
        (void)hipEventDestroy(ctx->lf_fork);
    if (ctx->lf_join)
        (void)hipEventDestroy(ctx->lf_join);
    if (ctx->lf_ready)
        (void)hipEventDestroy(ctx->lf_ready);
    if (ctx->peer_event)
        (void)hipEventDestroy(ctx->peer_event);
    if (ctx->peer_floor)
        (void)hipFree(ctx->peer_floor);
    if (ctx->verify_sums)
        (void)hipFree(ctx->verify_sums);
    if (ctx->h_lf_total_pinned)
        (void)hipHostFree(ctx->h_lf_total_pinned);
    void *lfdev[] = {ctx->lf_recs, ctx->lf_codes, ctx->lf_work, ctx->lf_streams, ctx->lf_bits, ctx->lf_packed, ctx->lf_total};
    for (void *p : lfdev)
        if (p)
            (void)hipFree(p);
    void *dev[] = {ctx->rans_aux, ctx->rans_flags, ctx->rans_final, ctx->rbits_total, ctx->tokens, ctx->bitbuf, ctx->tables, ctx->dc, ctx->accum, ctx->sym_count, ctx->part_info, ctx->group_bits,
                   ctx->offsets, ctx->total, ctx->d_jobs, /* the LUTs are the device's, shared by its contexts */
                   ctx->payload, ctx->dbg_xyb, ctx->dbg_dct, ctx->dbg_quant, ctx->d_arena};
    for (void *p : dev)
        if (p)
            (void)hipFree(p);
    for (int i = 0; i < kStaging; i++) {
        if (ctx->pinned[i])
            (void)hipHostFree(ctx->pinned[i]);
        if (ctx->staged[i])
            (void)hipEventDestroy(ctx->staged[i]);
    }
    for (int i = 0; i < 4; i++) {
        if (ctx->h_jobs_ring[i])
            (void)hipHostFree(ctx->h_jobs_ring[i]);
        if (ctx->jobs_uploaded[i])
            (void)hipEventDestroy(ctx->jobs_uploaded[i]);
    }
    if (ctx->h_total_pinned)
        (void)hipHostFree(ctx->h_total_pinned);
    if (ctx->h_status_pinned)
        (void)hipHostFree(ctx->h_status_pinned);
    if (ctx->own_stream)
        (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

/* hard upper bound of a frame's packed sections for the current token capacity: per symbol one 16-bit
 * refill + the longest residue (30 bits float, 13 integer), per group state + preset + padding */
/* HYDK_STATUS_TOKENS / _PAYLOAD: some context of this process ran a frame twice because that array's default size was too
 * small (resolve_overflow); hydamd_begin_frame lets idle contexts enlarge theirs ahead of their next frame */
static std::atomic<uint32_t> g_outgrown{0};

static size_t payload_bound(const HydAmdContext *ctx) {
    const size_t per_group = ((size_t)ctx->tok_cap * (ctx->rec_bytes == 8 ? 46 : 29) + 64 + 7) / 8;
    return (size_t)ctx->max_slots * HYDK_GROUPS_PER_LFG * per_group;
}

/* (re)allocate the arrays whose size follows tok_cap / rec_bytes / payload_cap; the stream must be idle */
static int alloc_frame_arrays(HydAmdContext *ctx, size_t payload_cap) {
    const size_t groups = (size_t)ctx->max_slots * HYDK_GROUPS_PER_LFG;
    void *old[] = {ctx->tokens, ctx->rans_aux, ctx->rans_flags, ctx->payload, ctx->bitbuf};
    for (void *p : old)
        if (p)
            (void)hipFree(p);
    ctx->tokens = nullptr;
    ctx->rans_aux = ctx->rans_flags = nullptr;
    ctx->payload = nullptr;
    const bool had_bitbuf = ctx->bitbuf != nullptr;
    ctx->bitbuf = nullptr;
    ctx->bit_pitch_words = 0;
    HIP_TRY(ctx, hipMalloc(&ctx->tokens, groups * ctx->tok_cap * ctx->rec_bytes));
    HIP_TRY(ctx, hipMalloc(&ctx->rans_aux, groups * ctx->tok_cap * sizeof(uint16_t)));
    HIP_TRY(ctx, hipMalloc(&ctx->rans_flags, groups * (ctx->tok_cap / 16) * sizeof(uint16_t)));
    ctx->payload_cap = payload_cap;
    HIP_TRY(ctx, hipMalloc(&ctx->payload, payload_cap + 16)); /* + 16: the emit and export kernels move whole words / 16-byte pieces */
    if (had_bitbuf) {
        ctx->bit_pitch_words = HYDK_BITWORDS_FOR(ctx->tok_cap);
        HIP_TRY(ctx, hipMalloc(&ctx->bitbuf, groups * ctx->bit_pitch_words * sizeof(uint32_t)));
    }
    return ST_OK;
}

/* the wave-per-group form's reversed bit buffers, on first use */
static int ensure_bitbuf(HydAmdContext *ctx) {
    if (ctx->bitbuf)
        return ST_OK;
    const size_t groups = (size_t)ctx->max_slots * HYDK_GROUPS_PER_LFG;
    ctx->bit_pitch_words = HYDK_BITWORDS_FOR(ctx->tok_cap);
    HIP_TRY(ctx, hipMalloc(&ctx->bitbuf, groups * ctx->bit_pitch_words * sizeof(uint32_t)));
    return ST_OK;
}

/* ---- per-device lookup tables (format.c:58-83), shared by all contexts ---- */
namespace {
struct SharedLuts {
    int device, linear_light;
    uint16_t *in_lut8, *in_lut16;
    float *bias_lut;
    int best_register_mode;
};
std::mutex g_luts_mutex;
std::vector<SharedLuts> g_luts; /* never freed: 385 KB per (device, transfer curve) for the life of the process */
} /* namespace */

static int acquire_shared_luts(HydAmdContext *ctx) {
    std::lock_guard<std::mutex> lock(g_luts_mutex);
    for (const SharedLuts &l : g_luts)
        if (l.device == ctx->device && l.linear_light == ctx->linear_light) {
            ctx->in_lut8 = l.in_lut8;
            ctx->in_lut16 = l.in_lut16;
            ctx->bias_lut = l.bias_lut;
            ctx->best_register_mode = l.best_register_mode;
            return ST_OK;
        }
    SharedLuts l = {ctx->device, ctx->linear_light, nullptr, nullptr, nullptr, 2};
    uint32_t *mism = nullptr;
    struct Guard { /* an error part of the way through hands back what was allocated so far */
        SharedLuts *l;
        uint32_t **mism;
        bool keep = false;
        ~Guard() {
            if (*mism)
                (void)hipFree(*mism);
            if (!keep) {
                if (l->in_lut8)
                    (void)hipFree(l->in_lut8);
                if (l->in_lut16)
                    (void)hipFree(l->in_lut16);
                if (l->bias_lut)
                    (void)hipFree(l->bias_lut);
            }
        }
    } guard{&l, &mism};
    HIP_TRY(ctx, hipMalloc(&l.in_lut8, 256 * sizeof(uint16_t)));
    HIP_TRY(ctx, hipMalloc(&l.in_lut16, 65536 * sizeof(uint16_t)));
    HIP_TRY(ctx, hipMalloc(&l.bias_lut, 65536 * sizeof(float)));
    std::vector<uint16_t> l8(256), l16(65536);
    std::vector<float> bias(65536);
    host_input_lut(l8.data(), 256, ctx->linear_light);
    host_input_lut(l16.data(), 65536, ctx->linear_light);
    const float unit = 1.0f / (65536 - 1.0f);
    for (size_t i = 0; i < 65536; i++)
        bias[i] = host_bias(i * unit);
    HIP_TRY(ctx, hipMemcpy(l.in_lut8, l8.data(), 256 * sizeof(uint16_t), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(l.in_lut16, l16.data(), 65536 * sizeof(uint16_t), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMemcpy(l.bias_lut, bias.data(), 65536 * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMalloc(&mism, sizeof(uint32_t)));
    for (int mode = 0; mode < 2 && l.best_register_mode == 2; mode++) {
        uint32_t h_mism = 1;
        HIP_TRY(ctx, hipMemsetAsync(mism, 0, sizeof(uint32_t), ctx->stream));
        HIP_TRY(ctx, hydk::launch_lut_selftest(l.in_lut16, l.bias_lut, ctx->linear_light, mode, mism, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(&h_mism, mism, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (h_mism == 0)
            l.best_register_mode = mode;
    }
    guard.keep = true;
    g_luts.push_back(l);
    ctx->in_lut8 = l.in_lut8;
    ctx->in_lut16 = l.in_lut16;
    ctx->bias_lut = l.bias_lut;
    ctx->best_register_mode = l.best_register_mode;
    return ST_OK;
}

static int create_impl(HydAmdContext *ctx, int debug_planes) {
    const size_t slots = (size_t)ctx->max_slots, G = HYDK_GROUPS_PER_LFG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    {
        /* HYDAMD_STREAM_HIGH=n (A/B knob): the first n contexts a process creates get their stream at the device's highest
         * priority — a staggered mix of stages instead of sixteen streams progressing alike (round 6; same bytes) */
        static std::atomic<int> created{0};
        static const int high = getenv("HYDAMD_STREAM_HIGH") ? atoi(getenv("HYDAMD_STREAM_HIGH")) : 0;
        int lo = 0, hi = 0;
        if (created.fetch_add(1) < high && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi != lo)
            HIP_TRY(ctx, hipStreamCreateWithPriority(&ctx->own_stream, hipStreamNonBlocking, hi));
        else
            HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
    }
    ctx->stream = ctx->own_stream;
    if (const char *env = getenv("HYDAMD_TOKEN_CAP")) { /* records per group before the overflow path kicks in (tests) */
        const long v = atol(env);
        if (v >= 16 && v <= HYDK_TOKENS_PER_GROUP)
            ctx->tok_cap = (uint32_t)v & ~15u;
        ctx->caps_forced = true;
    }
    {
        /* one byte per pixel of packed sections is 5 to 10 times what photographic content needs */
        size_t cap = slots * (size_t)2048 * 2048;
        if (const char *env = getenv("HYDAMD_PAYLOAD_CAP"))
            if (atol(env) > 0) {
                cap = (size_t)atol(env);
                ctx->caps_forced = true;
            }
        const size_t bound = payload_bound(ctx);
        const int st = alloc_frame_arrays(ctx, cap < bound ? cap : bound);
        if (st != ST_OK)
            return st;
    }
    HIP_TRY(ctx, hipMalloc(&ctx->rans_final, slots * G * sizeof(uint32_t)));
    HIP_TRY(ctx, hipMalloc(&ctx->rbits_total, slots * G * sizeof(uint32_t)));
    HIP_TRY(ctx, hipMalloc(&ctx->tables, slots * sizeof(HydkTables)));
    HIP_TRY(ctx, hipMalloc(&ctx->dc, slots * 3 * HYDK_DC_PITCH * HYDK_DC_PITCH * sizeof(int32_t)));
    {
        /* everything a frame ACCUMULATES into (histograms, alphabet maxima, status bits, the LF coder's
         * histograms) sits in one arena that k_frame_begin clears in the launch that stages the frame's
         * first job descriptors: one launch where four memsets and a copy used to sit in the stream */
        const size_t hist_words = slots * HYDK_MAX_CLUSTERS * HYDK_ALPHABET, lf_words = (slots + 1) * HYDK_LF_CODES; /* +1: the unit-test entry's scratch */
        const size_t head_words = (slots + 1 + 3) & ~(size_t)3; /* alpha_max[slots], status, padding to 16 bytes */
        ctx->accum_words = hist_words + head_words + ((lf_words + 3) & ~(size_t)3);
        HIP_TRY(ctx, hipMalloc(&ctx->accum, ctx->accum_words * sizeof(uint32_t)));
        HIP_TRY(ctx, hipMemset(ctx->accum, 0, ctx->accum_words * sizeof(uint32_t)));
        ctx->hist = ctx->accum;
        ctx->alpha_max = ctx->accum + hist_words;
        ctx->status = ctx->alpha_max + slots;
        ctx->lf_hist = ctx->accum + hist_words + head_words;
    }
    HIP_TRY(ctx, hipMalloc(&ctx->sym_count, slots * G * sizeof(uint32_t)));
    HIP_TRY(ctx, hipMalloc(&ctx->part_info, (size_t)(split_slots() > kSplitSlots ? split_slots() : kSplitSlots) * G * 4 * sizeof(uint2)));
    HIP_TRY(ctx, hipMalloc(&ctx->group_bits, slots * G * sizeof(uint32_t)));
    HIP_TRY(ctx, hipMalloc(&ctx->offsets, slots * G * sizeof(uint64_t)));
    HIP_TRY(ctx, hipMalloc(&ctx->total, sizeof(uint64_t)));
    HIP_TRY(ctx, hipMalloc(&ctx->d_jobs, slots * sizeof(HydkLfJob)));
    HIP_TRY(ctx, hipMalloc(&ctx->lf_recs, slots * HYDK_LF_SYMBOLS * sizeof(unsigned long long)));
    HIP_TRY(ctx, hipMalloc(&ctx->lf_codes, HYDK_LF_CODES * sizeof(uint32_t)));
    HIP_TRY(ctx, hipMalloc(&ctx->lf_work, slots * hydk::lf_work_bytes()));
    HIP_TRY(ctx, hipMalloc(&ctx->lf_streams, (slots + 1) * sizeof(HydkLfStream)));
    HIP_TRY(ctx, hipMalloc(&ctx->lf_bits, slots * HYDK_LF_BITWORDS * sizeof(uint32_t)));
    HIP_TRY(ctx, hipMalloc(&ctx->lf_packed, slots * HYDK_LF_BITWORDS * sizeof(uint32_t)));
    HIP_TRY(ctx, hipMalloc(&ctx->lf_total, sizeof(unsigned long long)));
    HIP_TRY(ctx, hipHostMalloc((void **)&ctx->h_lf_total_pinned, sizeof(unsigned long long), hipHostMallocDefault));
    *ctx->h_lf_total_pinned = 0;
    HIP_TRY(ctx, hipMemset(ctx->lf_streams, 0, (slots + 1) * sizeof(HydkLfStream)));
    /* lf_stream: created on first use (ensure_lf_stream).  A stream takes a place in the runtime's rotation over the
     * hardware queues whether it ever carries work or not: thirty-two contexts that code their LF groups in their own
     * stream would put their main streams on every second queue only */
    /* Events that only ORDER work on this device — the staging fence, the LF side stream's fork and join, the descriptor
     * ring's "this slot has been read" — could release at DEVICE scope (what the host does behind them needs no data the
     * GPU wrote): HYDAMD_EVENT_SCOPE=device, an A/B switch.  Measured equal in the pipelined loop (153.1 / 153.6 against
     * 153.1 / 153.6, profiles/r06_launch_boundaries.txt), so the runtime's default stays.  lf_ready is never touched: the
     * host reads the pinned LF total behind it. */
    static const unsigned order_only = [] {
        const char *v = getenv("HYDAMD_EVENT_SCOPE");
#ifdef HYD_TEST_HOOKS
        if (v && !strcmp(v, "device-all")) /* (see lf_ready) */
            return (unsigned)hipEventReleaseToDevice;
#endif
        return v && !strcmp(v, "device") ? (unsigned)hipEventReleaseToDevice : 0u;
    }();
    HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->frame_fence, hipEventDisableTiming | order_only));
    HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->lf_fork, hipEventDisableTiming | order_only));
    HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->lf_join, hipEventDisableTiming | order_only));
    {
#ifdef HYD_TEST_HOOKS
        /* (HYDAMD_EVENT_SCOPE=device-all, probe flavour only: lf_ready too — is its system-scope release what the LF coder costs
         * the loop?  It is not: profiles/r06_dyn_lds.txt) */
        const char *v = getenv("HYDAMD_EVENT_SCOPE");
        const unsigned all = v && !strcmp(v, "device-all") ? (unsigned)hipEventReleaseToDevice : 0u;
#else
        const unsigned all = 0u;
#endif
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->lf_ready, hipEventDisableTiming | all));
    }
    for (int i = 0; i < 4; i++) {
        HIP_TRY(ctx, hipHostMalloc((void **)&ctx->h_jobs_ring[i], slots * sizeof(HydkLfJob), hipHostMallocDefault));
        memset(ctx->h_jobs_ring[i], 0, slots * sizeof(HydkLfJob));
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->jobs_uploaded[i], hipEventDisableTiming | order_only));
    }
    ctx->h_jobs = ctx->h_jobs_ring[0];
    HIP_TRY(ctx, hipHostMalloc((void **)&ctx->h_total_pinned, sizeof(uint64_t), hipHostMallocDefault));
    HIP_TRY(ctx, hipHostMalloc((void **)&ctx->h_status_pinned, sizeof(uint32_t), hipHostMallocDefault));
    HIP_TRY(ctx, hipMemset(ctx->group_bits, 0, slots * G * sizeof(uint32_t)));
    HIP_TRY(ctx, hipMemset(ctx->sym_count, 0, slots * G * sizeof(uint32_t)));
    /* the memsets above run in the NULL stream and return before they have run; the context's own stream is
     * non-blocking and does not wait for that stream: wait here, once, or its first kernels may pass them */
    HIP_TRY(ctx, hipStreamSynchronize(nullptr));
    if (debug_planes) {
        HIP_TRY(ctx, hipMalloc(&ctx->dbg_xyb, 3 * kDbgPlane * sizeof(float)));
        HIP_TRY(ctx, hipMalloc(&ctx->dbg_dct, 3 * kDbgPlane * sizeof(float)));
        HIP_TRY(ctx, hipMalloc(&ctx->dbg_quant, 3 * kDbgPlane * sizeof(int32_t)));
    }

    /* LUTs: built on the host, uploaded ONCE per device and transfer curve and shared by every context of the process
     * (sixteen contexts' private copies were 6 MB of tables competing for each 4 MB L2 — the gathers of the transform
     * kernel then miss); the register evaluation is checked against them, also once */
    {
        const int st = acquire_shared_luts(ctx);
        if (st != ST_OK)
            return st;
    }
    ctx->register_luts_ok = ctx->best_register_mode < 2;
    ctx->use_luts = ctx->best_register_mode;
    if (const char *env = getenv("HYDAMD_RANS_WAVES")) {
        const int w = atoi(env);
        if (w >= 4 && w <= 6)
            ctx->rans_lanes = w - 4;
        else if (w >= 1 && w <= 3)
            ctx->rans_lanes = retired_rans_form(w);
    }
    if (const char *env = getenv("HYDAMD_CURVE_GATHERS")) /* A/B: 1 always, 2 never */
        if (atoi(env) >= 0 && atoi(env) <= 2)
            ctx->curve_gathers = atoi(env);
    if (const char *env = getenv("HYDAMD_LF_CODER")) /* 0: leave the LF ints to the host coder (A/B measurements) */
        ctx->lf_on_device = atoi(env) == 2 ? 2 : atoi(env) != 0;
    if (const char *env = getenv("HYDAMD_XYB_MODE")) { /* 0 / 1 / 2, never faster than what was proven exact */
        const int m = atoi(env);
        if (m >= ctx->best_register_mode && m <= 2)
            ctx->use_luts = m;
    }
    return ST_OK;
}

HydAmdContext *hydamd_create(int device, int max_lf_groups, int linear_light, int debug_planes, int *status) {
    int st = ST_OK;
    HydAmdContext *ctx = nullptr;
    if (max_lf_groups < 1 || max_lf_groups > HYDAMD_MAX_LF_GROUPS) {
        st = fail(nullptr, ST_API_ERROR, "max_lf_groups must be between 1 and 255");
    } else if (hydamd_device_count() <= device || device < 0) {
        st = fail(nullptr, ST_INTERNAL_ERROR, "no usable HIP device (this build has no CPU fallback)");
    } else {
        ctx = new (std::nothrow) HydAmdContext();
        if (!ctx) {
            st = fail(nullptr, ST_NOMEM, "out of host memory");
        } else {
            ctx->device = device;
            ctx->max_slots = max_lf_groups;
            ctx->linear_light = linear_light != 0;
            st = create_impl(ctx, debug_planes);
            if (st != ST_OK) {
                snprintf(g_global_error, sizeof(g_global_error), "%s", ctx->error);
                hydamd_destroy(ctx);
                ctx = nullptr;
            }
        }
    }
    if (status)
        *status = st;
    return ctx;
}

int hydamd_set_stream(HydAmdContext *ctx, void *hip_stream) {
    if (!ctx)
        return ST_API_ERROR;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return ST_OK;
}

void *hydamd_get_stream(HydAmdContext *ctx) { return ctx ? (void *)ctx->stream : nullptr; }


int hydamd_uses_register_luts(HydAmdContext *ctx) { return ctx && ctx->use_luts < 2; }

int hydamd_xyb_mode(HydAmdContext *ctx) { return ctx ? ctx->use_luts : -1; }

/* an LF group's descriptor names the transform kernel variant of the mode it was RECORDED under (job.use_luts); the launch
 * picks its variant from the context's mode at launch time: a mode change in between would leave the group to no kernel */
static bool recorded_but_not_transformed(const HydAmdContext *ctx) {
    for (int i = ctx->transformed; i < ctx->max_slots; i++)
        if (ctx->h_jobs[i].width)
            return true;
    return false;
}

int hydamd_force_luts(HydAmdContext *ctx, int use_luts) {
    if (!ctx)
        return ST_API_ERROR;
    if (recorded_but_not_transformed(ctx))
        return fail(ctx, ST_API_ERROR, "the XYB mode cannot change between recording an LF group and its transform stage");
    if (!use_luts && !ctx->register_luts_ok)
        return fail(ctx, ST_INTERNAL_ERROR, "register LUT evaluation failed its self-test on this device");
    ctx->use_luts = use_luts ? 2 : ctx->best_register_mode;
    return ST_OK;
}

int hydamd_set_xyb_mode(HydAmdContext *ctx, int mode) {
    if (!ctx)
        return ST_API_ERROR;
    if (recorded_but_not_transformed(ctx))
        return fail(ctx, ST_API_ERROR, "the XYB mode cannot change between recording an LF group and its transform stage");
    if (mode < ctx->best_register_mode || mode > 2)
        return fail(ctx, ST_API_ERROR, "XYB mode not available (it must have passed the bit-exactness self-test)");
    ctx->use_luts = mode;
    return ST_OK;
}

int hydamd_forget_content(HydAmdContext *ctx) {
    if (!ctx)
        return ST_API_ERROR;
    ctx->seen_bytes = ctx->seen_pixels = 0;
    return ST_OK;
}

int hydamd_set_curve_gathers(HydAmdContext *ctx, int mode) {
    if (!ctx)
        return ST_API_ERROR;
    if (mode < 0 || mode > 2)
        return fail(ctx, ST_API_ERROR, "curve gathers: 0 by content, 1 always, 2 never");
    ctx->curve_gathers = mode;
    return ST_OK;
}

int hydamd_set_rans_waves(HydAmdContext *ctx, int waves) {
    if (!ctx)
        return ST_API_ERROR;
    if (waves >= 1 && waves <= 3) { /* round 1's row forms: callers that still name one get their successor */
        ctx->rans_lanes = retired_rans_form(waves);
        return ST_OK;
    }
    if (waves < 4 || waves > 6)
        return fail(ctx, ST_API_ERROR, "entropy-stage form must be 4 (wave per group) or 5 (lane per group; 6 is accepted as a synonym)");
    ctx->rans_lanes = waves - 4;
    return ST_OK;
}

int hydamd_begin_frame(HydAmdContext *ctx, unsigned num_presets) {
    if (!ctx)
        return ST_API_ERROR;
    /* 128 or 256 presets would need 256 clusters, which the reference cannot code (entropy.c:99) */
    if (num_presets < 1 || num_presets > HYDAMD_MAX_LF_GROUPS || num_presets == 128)
        return fail(ctx, ST_API_ERROR, "unsupported number of LF groups per frame (1..255 except 128)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->num_presets = num_presets;
    /* clustering scheme by preset count (encoder.c:862-901) */
    if (num_presets * 9 <= 256) {
        ctx->scheme = 0;
        ctx->nclusters = 9;
    } else if (num_presets * 3 <= 256) {
        ctx->scheme = 1;
        ctx->nclusters = 3;
    } else if (num_presets * 2 <= 256) {
        ctx->scheme = 2;
        ctx->nclusters = 2;
    } else {
        ctx->scheme = 3;
        ctx->nclusters = 1;
    }
    int bits = 0;
    while ((1u << bits) < num_presets)
        bits++;
    ctx->preset_bits = bits; /* hyd_cllog2(num_presets), encoder.c:940 */
    {
        /* Another context of this process met content that outgrew the default arrays (noise: 2.9 symbols and 1.8 section
         * bytes per pixel) and had to run its frame twice.  A queue of frames is of one kind as a rule: a context whose
         * stream is idle right now takes the hard maxima BEFORE its frame instead of finding out by itself (busy: it keeps
         * what it has — never a wait here; sizes forced through HYDAMD_TOKEN_CAP / HYDAMD_PAYLOAD_CAP stay as forced). */
        const uint32_t seen = g_outgrown.load(std::memory_order_relaxed);
        const bool tok = (seen & HYDK_STATUS_TOKENS) && ctx->tok_cap < HYDK_TOKENS_PER_GROUP;
        const bool pay = (seen & HYDK_STATUS_PAYLOAD) && ctx->payload_cap < payload_bound(ctx);
        if ((tok || pay) && !ctx->caps_forced && hipStreamQuery(ctx->stream) == hipSuccess &&
            (!ctx->lf_stream || hipStreamQuery(ctx->lf_stream) == hipSuccess)) {
            if (tok)
                ctx->tok_cap = HYDK_TOKENS_PER_GROUP;
            const int st = alloc_frame_arrays(ctx, pay ? payload_bound(ctx) : ctx->payload_cap);
            if (st != ST_OK)
                return st;
            ctx->grown_ahead++;
        }
        (void)hipGetLastError(); /* (hipErrorNotReady of a busy stream is not an error) */
    }
    ctx->results_valid = false;
    ctx->slots_finished = 0;
    ctx->slots_per_frame = 0;
    ctx->transformed = ctx->coded = ctx->lf_coded = 0;
    ctx->want_transform = ctx->want_entropy = 0;
    ctx->host_staged = 0;
    ctx->lf_need_gather = false;
    ctx->lf_results_valid = false;
    ctx->lf_slots = 0;
    /* this frame's uploads may overwrite the staging arena only after everything queued so far has read it */
    HIP_TRY(ctx, hipEventRecord(ctx->frame_fence, ctx->stream));
    ctx->copy_needs_fence = true;
    /* job descriptors are uploaded asynchronously from a small pinned ring, so up to three frames
     * can be queued behind the one executing without the host touching a descriptor in flight */
    ctx->jobs_idx = (int)(ctx->frame_counter++ & 3u);
    HIP_TRY(ctx, hipEventSynchronize(ctx->jobs_uploaded[ctx->jobs_idx]));
    ctx->h_jobs = ctx->h_jobs_ring[ctx->jobs_idx];
    memset(ctx->h_jobs, 0, (size_t)ctx->max_slots * sizeof(HydkLfJob));
    ctx->accum_stale = true; /* cleared by the launch that stages this frame's first job descriptors */
    ctx->alpha_floor = 0;
    ctx->alpha_floor_dev = nullptr;
    return ST_OK;
}

int hydamd_encode_lf_group(HydAmdContext *ctx, int slot, const void *const src[3], ptrdiff_t row_stride,
                           ptrdiff_t pixel_stride, int sample_fmt, size_t width, size_t height, unsigned preset) {
    int st = check_slot(ctx, slot);
    if (st != ST_OK)
        return st;
    if (!src || !src[0] || !src[1] || !src[2])
        return fail(ctx, ST_API_ERROR, "null pixel pointer");
    return record_lf_group(ctx, slot, src, row_stride, pixel_stride, sample_fmt, width, height, preset);
}

int hydamd_encode_lf_group_host(HydAmdContext *ctx, int slot, const void *const src[3], ptrdiff_t row_stride,
                                ptrdiff_t pixel_stride, int sample_fmt, size_t width, size_t height, unsigned preset) {
    int st = check_slot(ctx, slot);
    if (st != ST_OK)
        return st;
    if (!src || !src[0] || !src[1] || !src[2])
        return fail(ctx, ST_API_ERROR, "null pixel pointer");
    if (width == 0 || height == 0 || width > 2048 || height > 2048)
        return fail(ctx, ST_API_ERROR, "LF group must be between 1 and 2048 pixels in each direction");
    if (sample_fmt != HYDK_FMT_U8 && sample_fmt != HYDK_FMT_U16 && sample_fmt != HYDK_FMT_F32)
        return fail(ctx, ST_API_ERROR, "Invalid Sample Format");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t ss = sample_size(sample_fmt);
    const size_t bytes = width * height * 3 * ss;
    /* size for full 2048x2048 tiles of this sample type so that later tiles never reallocate */
    st = ensure_staging(ctx, (size_t)2048 * 2048 * 3 * ss);
    if (st != ST_OK)
        return st;
    const int k = ctx->staging_next;
    ctx->staging_next = (k + 1) % kStaging;
    HIP_TRY(ctx, hipEventSynchronize(ctx->staged[k])); /* previous upload from this pinned tile has left */
    if (sample_fmt == HYDK_FMT_U8)
        gather_packed((uint8_t *)ctx->pinned[k], src, row_stride, pixel_stride, width, height);
    else if (sample_fmt == HYDK_FMT_U16)
        gather_packed((uint16_t *)ctx->pinned[k], src, row_stride, pixel_stride, width, height);
    else
        gather_packed((float *)ctx->pinned[k], src, row_stride, pixel_stride, width, height);
    char *base = ctx->d_arena + (size_t)slot * ctx->arena_tile;
    if (ctx->copy_needs_fence) {
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->frame_fence, 0));
        ctx->copy_needs_fence = false;
    }
    HIP_TRY(ctx, hipMemcpyAsync(base, ctx->pinned[k], bytes, hipMemcpyHostToDevice, ctx->copy_stream));
    HIP_TRY(ctx, hipEventRecord(ctx->staged[k], ctx->copy_stream));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->staged[k], 0)); /* kernels queued from here on see the tile */
    const void *dsrc[3] = {base, base + ss, base + 2 * ss};
    st = record_lf_group(ctx, slot, dsrc, (ptrdiff_t)(3 * width), 3, sample_fmt, width, height, preset);
    if (st == ST_OK && slot + 1 > ctx->host_staged)
        ctx->host_staged = slot + 1;
    return st;
}

/* One call for a whole device-resident frame: begin, every LF group in raster order (slot = raster index = preset,
 * the one-frame layout of reference libhydrium.c:172-203 for tiles sent in raster order), the closing stage.  What a
 * caller with frames in HBM wants, and what keeps a host loop out of the frame rate: eighteen calls through a binding
 * (Python: 0.44 ms of host time per 8K frame — as much as the GPU needs for the frame) become one. */
int hydamd_encode_image(HydAmdContext *ctx, const void *const src[3], ptrdiff_t row_stride, ptrdiff_t pixel_stride,
                        int sample_fmt, size_t width, size_t height) {
    if (!ctx)
        return ST_API_ERROR;
    if (!src || !src[0] || !src[1] || !src[2])
        return fail(ctx, ST_API_ERROR, "null pixel pointer");
    if (width == 0 || height == 0)
        return fail(ctx, ST_API_ERROR, "empty image");
    const size_t lfx = (width + 2047) >> 11, lfy = (height + 2047) >> 11;
    if (lfx * lfy > (size_t)ctx->max_slots)
        return fail(ctx, ST_API_ERROR, "context has too few LF-group slots for this image");
    if (sample_fmt != HYDK_FMT_U8 && sample_fmt != HYDK_FMT_U16 && sample_fmt != HYDK_FMT_F32)
        return fail(ctx, ST_API_ERROR, "Invalid Sample Format");
    int st = hydamd_begin_frame(ctx, (unsigned)(lfx * lfy));
    const ptrdiff_t ss = (ptrdiff_t)sample_size(sample_fmt);
    for (size_t ty = 0; ty < lfy && st == ST_OK; ty++)
        for (size_t tx = 0; tx < lfx && st == ST_OK; tx++) {
            const ptrdiff_t off = ((ptrdiff_t)(ty * 2048) * row_stride + (ptrdiff_t)(tx * 2048) * pixel_stride) * ss;
            const void *p[3] = {(const char *)src[0] + off, (const char *)src[1] + off, (const char *)src[2] + off};
            const size_t w = width - tx * 2048 < 2048 ? width - tx * 2048 : 2048;
            const size_t h = height - ty * 2048 < 2048 ? height - ty * 2048 : 2048;
            st = hydamd_encode_lf_group(ctx, (int)(ty * lfx + tx), p, row_stride, pixel_stride, sample_fmt, w, h,
                                        (unsigned)(ty * lfx + tx));
        }
    if (st == ST_OK)
        st = hydamd_finish_frame(ctx, (int)(lfx * lfy));
    return st;
}

/* A BATCH of independent frames of one shape as one launch group: frame k occupies slots k * num_presets ... (preset ids
 * restart with every frame, and so does the running alphabet maximum of the table kernel), every kernel of the closing
 * stage covers all of them.  The serial rANS chains of the frames then run side by side: a stream is held for one chain's
 * 2.5 ms per BATCH instead of per frame — the lever on the frame rate of a queue of frames.  The sections of
 * frame k follow those of frame k - 1 in the payload (hydamd_read_sections per slot), likewise the packed LF streams. */
int hydamd_begin_batch(HydAmdContext *ctx, unsigned num_presets, int frames) {
    if (!ctx)
        return ST_API_ERROR;
    if (frames < 1 || (size_t)frames * num_presets > (size_t)ctx->max_slots)
        return fail(ctx, ST_API_ERROR, "the context has too few LF-group slots for this batch");
    const int st = hydamd_begin_frame(ctx, num_presets);
    if (st == ST_OK && frames > 1)
        ctx->slots_per_frame = (int)num_presets;
    return st;
}

int hydamd_encode_image_batch(HydAmdContext *ctx, int frames, const void *const *src /* [frames][3] */, ptrdiff_t row_stride,
                              ptrdiff_t pixel_stride, int sample_fmt, size_t width, size_t height) {
    if (!ctx)
        return ST_API_ERROR;
    if (!src || frames < 1)
        return fail(ctx, ST_API_ERROR, "null pixel pointer");
    if (width == 0 || height == 0)
        return fail(ctx, ST_API_ERROR, "empty image");
    if (sample_fmt != HYDK_FMT_U8 && sample_fmt != HYDK_FMT_U16 && sample_fmt != HYDK_FMT_F32)
        return fail(ctx, ST_API_ERROR, "Invalid Sample Format");
    const size_t lfx = (width + 2047) >> 11, lfy = (height + 2047) >> 11, n = lfx * lfy;
    int st = hydamd_begin_batch(ctx, (unsigned)n, frames);
    const ptrdiff_t ss = (ptrdiff_t)sample_size(sample_fmt);
    for (int f = 0; f < frames && st == ST_OK; f++) {
        const void *const *s3 = src + 3 * (size_t)f;
        if (!s3[0] || !s3[1] || !s3[2])
            return fail(ctx, ST_API_ERROR, "null pixel pointer");
        for (size_t ty = 0; ty < lfy && st == ST_OK; ty++)
            for (size_t tx = 0; tx < lfx && st == ST_OK; tx++) {
                const ptrdiff_t off = ((ptrdiff_t)(ty * 2048) * row_stride + (ptrdiff_t)(tx * 2048) * pixel_stride) * ss;
                const void *p[3] = {(const char *)s3[0] + off, (const char *)s3[1] + off, (const char *)s3[2] + off};
                const size_t w = width - tx * 2048 < 2048 ? width - tx * 2048 : 2048;
                const size_t h = height - ty * 2048 < 2048 ? height - ty * 2048 : 2048;
                st = hydamd_encode_lf_group(ctx, (int)((size_t)f * n + ty * lfx + tx), p, row_stride, pixel_stride, sample_fmt, w, h,
                                            (unsigned)(ty * lfx + tx));
            }
    }
    if (st == ST_OK)
        st = hydamd_finish_frame(ctx, (int)((size_t)frames * n));
    return st;
}

/* K1 for slots [first, first + count) */
static int transform_range(HydAmdContext *ctx, int first, int count) {
    unsigned mask = 0;
    for (int i = first; i < first + count; i++) {
        if (ctx->h_jobs[i].width == 0)
            return fail(ctx, ST_API_ERROR, "an LF-group slot of this frame was never submitted");
        mask |= 1u << ctx->h_jobs[i].fmt;
    }
    ctx->status_published = false;
    /* the descriptors travel by a kernel that reads the pinned ring; the frame's first such launch also
     * clears what the frame accumulates into */
    HIP_TRY(ctx, hydk::launch_frame_begin(ctx->h_jobs + first, ctx->d_jobs + first, count,
                                          ctx->accum_stale ? ctx->accum : nullptr, ctx->accum_words, ctx->stream));
    ctx->accum_stale = false;
    HIP_TRY(ctx, hipEventRecord(ctx->jobs_uploaded[ctx->jobs_idx], ctx->stream));
    ScopedTimer timer(ctx, HYDAMD_K_TRANSFORM);
    /* One of a pixel's six curves comes from the uploaded table through the texture path (HYDK_K1_GATHER: photo -9 %,
     * smooth -8 %) — unless the picture's pixels scatter over the whole table, where every lane's gather is its own L2
     * miss (random noise +12 %).  No one tells the encoder what it is about to see; what it has is the last frame of this
     * context that hydamd_sync waited for — section bytes and pixel count snapshotted TOGETHER there (until round 5 the pinned
     * total was read here unsynchronised, possibly a frame behind, against a pixel count that could belong to another frame:
     * bit-exact either way, but which kernel ran was timing-dependent): above 0.75 bytes per pixel — photographic content
     * has 0.15, noise 2 — the next transform kernels evaluate all six curves in registers (modes 3, 4).  A loop that never
     * waits between frames keeps the choice of its last wait; hydamd_forget_content clears it (a parked context reused for
     * another image). */
    int xmode = ctx->use_luts;
    if (xmode < 2) {
        bool dense = ctx->curve_gathers == 2;
        if (ctx->curve_gathers == 0 && ctx->seen_pixels)
            dense = (double)ctx->seen_bytes > 0.75 * (double)ctx->seen_pixels;
        if (dense)
            xmode += 3;
    }
    /* a launch of one or two LF groups (a tile-mode frame, a tile of the drop-in API's one-frame mode) is 64 or 128
     * workgroups on 256 compute units, each walking its group's 32 strips one after another: 0.19 ms whatever the tile.
     * Such launches split every group over four workgroups (eight strips each; k_join_parts closes the token array up):
     * HYDAMD_K1_SPLIT=0 for A/B */
    static const bool split_on = !(getenv("HYDAMD_K1_SPLIT") && atoi(getenv("HYDAMD_K1_SPLIT")) == 0);
    static const int split_log = getenv("HYDAMD_K1_SPLIT_LOG") && atoi(getenv("HYDAMD_K1_SPLIT_LOG")) == 1 ? 1 : 2;
    const int plog = split_on && count <= split_slots() && ctx->tok_cap % 64 == 0 ? split_log : 0;
    HIP_TRY(ctx, hydk::launch_transform(ctx->d_jobs + first, count, mask, xmode, ctx->status, ctx->part_info, plog, ctx->stream));
    return ST_OK;
}

/* The LF coder's token and code kernels for slots [first, first + count), whose transform kernels
 * are already enqueued.  It needs only the LF ints they write: forked onto its own stream so that it
 * overlaps the HF entropy stage; join_lf brings the streams back together. */
#ifdef HYD_TEST_HOOKS
/* HYDAMD_DEBUG_LF_DROP (timing only, wrong LF streams): 1 the token kernel, 2 the code passengers, 4 offsets + pack, 8 the gather */
static int lf_drop() {
    static const int v = getenv("HYDAMD_DEBUG_LF_DROP") ? atoi(getenv("HYDAMD_DEBUG_LF_DROP")) : 0;
    return v;
}
#else
static constexpr int lf_drop() { return 0; }
#endif
static int ensure_lf_stream(HydAmdContext *ctx) {
    if (!ctx->lf_stream)
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->lf_stream, hipStreamNonBlocking));
    return ST_OK;
}

static int lf_range(HydAmdContext *ctx, int first, int count, bool forked) {
    if (forked) {
        const int st = ensure_lf_stream(ctx);
        if (st != ST_OK)
            return st;
    }
    hipStream_t where = forked ? ctx->lf_stream : ctx->stream;
    ctx->status_published = false;
    if (forked) {
        HIP_TRY(ctx, hipEventRecord(ctx->lf_fork, ctx->stream));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->lf_stream, ctx->lf_fork, 0));
    }
    ScopedTimer timer(ctx, HYDAMD_K_LF, where);
    HIP_TRY(ctx, hydk::launch_lf_coder(ctx->d_jobs + first, ctx->lf_recs + (size_t)first * HYDK_LF_SYMBOLS,
                                       ctx->lf_hist + (size_t)first * HYDK_LF_CODES, ctx->lf_streams + first,
                                       ctx->lf_bits + (size_t)first * HYDK_LF_BITWORDS,
                                       ctx->lf_work + (size_t)first * hydk::lf_work_bytes(), count, where));
    ctx->lf_pending = ctx->lf_pending || forked;
    ctx->lf_need_gather = true;
    return ST_OK;
}

int hydamd_run_transform(HydAmdContext *ctx, int num_slots) {
    if (!ctx)
        return ST_API_ERROR;
    if (num_slots < 1 || num_slots > ctx->max_slots)
        return fail(ctx, ST_API_ERROR, "slot count out of range");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->results_valid = false;
    if (num_slots > ctx->want_transform)
        ctx->want_transform = num_slots;
    if (num_slots > ctx->transformed) {
        const int st = transform_range(ctx, ctx->transformed, num_slots - ctx->transformed);
        if (st != ST_OK)
            return st;
        ctx->transformed = num_slots;
    }
    if (ctx->lf_on_device == 1 && num_slots > ctx->lf_coded) {
        const int st = lf_range(ctx, ctx->lf_coded, num_slots - ctx->lf_coded, true);
        if (st != ST_OK)
            return st;
        ctx->lf_coded = num_slots;
    }
    return ST_OK;
}

/* the frame's LF streams are packed once all of its LF groups are coded (num_slots > 0: called from
 * the frame's closing stage); then the side stream rejoins the main one */
static int join_lf(HydAmdContext *ctx, int num_slots, bool totals_follow = false) {
    if (num_slots > 0 && ctx->lf_on_device == 2 && num_slots > ctx->lf_coded) {
        /* in-stream mode: the whole LF coder runs here, behind the frame's packing kernels (no extra
         * stream: with many frames in flight side streams alias onto the same hardware queues) */
        const int st = lf_range(ctx, ctx->lf_coded, num_slots - ctx->lf_coded, false);
        if (st != ST_OK)
            return st;
        ctx->lf_coded = num_slots;
    }
    if (num_slots > 0 && ctx->lf_need_gather) {
        hipStream_t where = ctx->lf_pending ? ctx->lf_stream : ctx->stream;
        if (!(lf_drop() & 8))
        HIP_TRY(ctx, hydk::launch_lf_gather(ctx->lf_streams, ctx->lf_bits, ctx->lf_packed, ctx->lf_total, num_slots, where));
        if (totals_follow && !ctx->lf_pending) /* the caller's own publishing launch follows on the same stream */
            ctx->lf_total_unpublished = true;
        else /* the host may wait for the LF streams alone (lf_ready) */
            HIP_TRY(ctx, hydk::launch_publish(nullptr, nullptr, ctx->lf_total, ctx->h_lf_total_pinned, nullptr, nullptr, where));
        HIP_TRY(ctx, hipEventRecord(ctx->lf_ready, where));
        ctx->lf_slots = num_slots;
        ctx->lf_need_gather = false;
    }
    if (ctx->lf_pending) {
        HIP_TRY(ctx, hipEventRecord(ctx->lf_join, ctx->lf_stream));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->lf_join, 0));
        ctx->lf_pending = false;
    }
    return ST_OK;
}

/* Start the current frame over after its arrays were re-laid out: every recorded slot is re-bound,
 * what the transform kernels accumulate is cleared, and the stages the caller had asked for are
 * enqueued again (the job descriptors are still in the pinned ring; pixels are borrowed until sync). */
static int replay_frame(HydAmdContext *ctx) {
    const int recorded = ctx->want_transform > ctx->transformed ? ctx->want_transform : ctx->transformed;
    for (int i = 0; i < ctx->max_slots; i++)
        if (ctx->h_jobs[i].width)
            bind_slot_buffers(ctx, i);
    ctx->accum_stale = true;
    const int entropy = ctx->want_entropy;
    ctx->transformed = ctx->coded = ctx->lf_coded = 0;
    ctx->results_valid = false;
    if (recorded > 0) {
        const int st = hydamd_run_transform(ctx, recorded);
        if (st != ST_OK)
            return st;
    }
    if (entropy > 0)
        return hydamd_run_entropy(ctx, entropy);
    return ST_OK;
}

/* The stream is idle and `status` holds the frame's status word.  If a buffer was too small: enlarge
 * it to its hard maximum and run the frame again.  Returns ST_OK with *again set when the caller
 * has to wait for the stream once more. */
static int resolve_overflow(HydAmdContext *ctx, uint32_t status, bool *again) {
    *again = false;
    if (!(status & HYDK_STATUS_OVERFLOW))
        return ST_OK;
    if (status & HYDK_STATUS_LAYOUT)
        return fail(ctx, ST_INTERNAL_ERROR, "float LF group in a context laid out for integer token records");
    /* the LF coder forked onto its side stream may still be reading the LF ints and adding into the
     * histograms the replay clears and rewrites: it has to be done before anything is re-laid or rerun */
    if (ctx->lf_stream)
        HIP_TRY(ctx, hipStreamSynchronize(ctx->lf_stream));
    size_t payload_cap = ctx->payload_cap;
    if (status & HYDK_STATUS_TOKENS) {
        if (ctx->tok_cap >= HYDK_TOKENS_PER_GROUP)
            return fail(ctx, ST_INTERNAL_ERROR, "a group produced more symbols than the format allows");
        ctx->tok_cap = HYDK_TOKENS_PER_GROUP;
    }
    if (status & HYDK_STATUS_PAYLOAD) {
        if (payload_cap >= payload_bound(ctx))
            return fail(ctx, ST_INTERNAL_ERROR, "sections larger than their hard bound");
        payload_cap = payload_bound(ctx);
    }
    int st = alloc_frame_arrays(ctx, payload_cap < payload_bound(ctx) ? payload_cap : payload_bound(ctx));
    if (st != ST_OK)
        return st;
    ctx->overflow_reruns++;
    if (!ctx->caps_forced) /* the process's other contexts take the hint at their next frame (hydamd_begin_frame) */
        g_outgrown.fetch_or(status & (HYDK_STATUS_TOKENS | HYDK_STATUS_PAYLOAD), std::memory_order_relaxed);
    st = replay_frame(ctx);
    *again = st == ST_OK;
    return st;
}

/* float input needs 8-byte token records: re-lay the token arrays once (the context stays wide) and
 * re-run the transform kernels already enqueued for this frame */
namespace {
int widen_token_records(HydAmdContext *ctx) {
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->lf_stream)
        HIP_TRY(ctx, hipStreamSynchronize(ctx->lf_stream));
    ctx->rec_bytes = 8;
    const size_t bound = payload_bound(ctx);
    int st = alloc_frame_arrays(ctx, ctx->payload_cap < bound ? ctx->payload_cap : bound);
    if (st != ST_OK)
        return st;
    if (ctx->transformed > 0 || ctx->want_entropy > 0)
        return replay_frame(ctx);
    for (int i = 0; i < ctx->max_slots; i++)
        if (ctx->h_jobs[i].width)
            bind_slot_buffers(ctx, i);
    return ST_OK;
}
} // namespace

/* wait for the stream; rerun the frame if one of its buffers turned out too small */
static int wait_for_frame(HydAmdContext *ctx) {
    for (int attempt = 0; attempt < 4; attempt++) {
        if (!ctx->status_published) /* stages were enqueued behind the frame's last publishing launch (or there was none) */
            HIP_TRY(ctx, hydk::launch_publish(nullptr, nullptr, nullptr, nullptr, ctx->status, ctx->h_status_pinned, ctx->stream));
        ctx->status_published = true;
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        bool again = false;
        const int st = resolve_overflow(ctx, *ctx->h_status_pinned, &again);
        if (st != ST_OK)
            return st;
        if (!again)
            return ST_OK;
        const int st2 = join_lf(ctx, 0);
        if (st2 != ST_OK)
            return st2;
    }
    return fail(ctx, ST_INTERNAL_ERROR, "frame still does not fit after enlarging its buffers");
}

int hydamd_read_alphabet_max(HydAmdContext *ctx, int slot, uint32_t *max_token_plus_one) {
    int st = check_slot(ctx, slot);
    if (st != ST_OK)
        return st;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    st = wait_for_frame(ctx); /* the maxima of a frame that outgrew its token arrays are incomplete */
    if (st != ST_OK)
        return st;
    HIP_TRY(ctx, hipMemcpy(max_token_plus_one, ctx->alpha_max + slot, sizeof(uint32_t), hipMemcpyDeviceToHost));
    return ST_OK;
}

int hydamd_set_alphabet_floor(HydAmdContext *ctx, uint32_t floor) {
    if (!ctx)
        return ST_API_ERROR;
    ctx->alpha_floor = floor;
    return ST_OK;
}

const uint32_t *hydamd_alphabet_max_device(HydAmdContext *ctx) { return ctx ? ctx->alpha_max : nullptr; }

int hydamd_set_alphabet_floor_device(HydAmdContext *ctx, const uint32_t *floor_on_device) {
    if (!ctx)
        return ST_API_ERROR;
    ctx->alpha_floor_dev = floor_on_device;
    return ST_OK;
}

/* ---- one frame on several devices, inside one process: no collective library, peer reads over xGMI ----
 * The reference codes a frame's LF groups one after another on one core (encoder.c:928-957); here every device owns a
 * run of consecutive LF groups (in send order) on a context of its own.  Two things cross devices: the running alphabet
 * maximum (one word per LF group, read from the earlier devices' contexts by a single-wave kernel) and, at the end, the
 * blobs, which the assembling device's kernels read in place through peer access. */
int hydamd_context_device(HydAmdContext *ctx) { return ctx ? ctx->device : -1; }

static int enable_peer_reads(HydAmdContext *ctx, int peer_device) {
    if (peer_device == ctx->device)
        return ST_OK; /* several contexts of one device (also how the tests alias a device list) */
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int can = 0;
    HIP_TRY(ctx, hipDeviceCanAccessPeer(&can, ctx->device, peer_device));
    if (!can)
        return fail(ctx, ST_INTERNAL_ERROR, "no peer access between two devices of HYDAMD_DEVICES");
    const hipError_t e = hipDeviceEnablePeerAccess(peer_device, 0);
    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
        return fail(ctx, ST_INTERNAL_ERROR, "hipDeviceEnablePeerAccess", e);
    (void)hipGetLastError();
    return ST_OK;
}

/* ctx's stream waits (on the device) for everything enqueued so far on peer's stream, and may read peer's memory */
int hydamd_wait_for(HydAmdContext *ctx, HydAmdContext *peer) {
    if (!ctx || !peer)
        return ST_API_ERROR;
    if (ctx == peer)
        return ST_OK;
    HIP_TRY(peer, hipSetDevice(peer->device));
    if (!peer->peer_event)
        /* what the waiting device reads is this device's memory, over xGMI: the record must release to SYSTEM scope (the
         * runtime's default for an event is device scope; this device's L2 slices are written back either way, its
         * memory-side cache is coherent for the fabric — the flag makes the intent explicit and costs nothing measurable) */
        HIP_TRY(peer, hipEventCreateWithFlags(&peer->peer_event, hipEventDisableTiming | hipEventReleaseToSystem));
    HIP_TRY(peer, hipEventRecord(peer->peer_event, peer->stream));
    const int st = enable_peer_reads(ctx, peer->device);
    if (st != ST_OK)
        return st;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, peer->peer_event, 0));
    return ST_OK;
}

namespace {
struct PeerMaxima {
    const uint32_t *max[HYDAMD_MAX_PEERS];
    int slots[HYDAMD_MAX_PEERS];
    int count;
};
__global__ __launch_bounds__(64) void k_floor_from_peers(PeerMaxima a, uint32_t *floor) {
    uint32_t m = 0;
    for (int p = 0; p < a.count; p++)
        for (int i = threadIdx.x; i < a.slots[p]; i += 64)
            m = max(m, a.max[p][i]);
#pragma unroll
    for (int d = 32; d; d >>= 1)
        m = max(m, (uint32_t)__shfl_xor((int)m, d));
    if (threadIdx.x == 0)
        *floor = m;
}
} /* namespace */

/* The floor of ctx's LF groups = the largest token + 1 over the LF groups sent before them, which the `npeers` contexts in
 * `peers` transformed (their transform stages must be enqueued: hydamd_run_transform or hydamd_submit_lf_group for every
 * slot): read from their memory by a kernel in ctx's stream, behind their transform kernels, and left where ctx's table
 * kernel looks for it.  No host synchronisation. */
int hydamd_alphabet_floor_from_peers(HydAmdContext *ctx, int npeers, HydAmdContext *const *peers) {
    if (!ctx || npeers < 0 || npeers > HYDAMD_MAX_PEERS || (npeers && !peers))
        return ctx ? fail(ctx, ST_API_ERROR, "bad peer list") : ST_API_ERROR;
    if (npeers == 0) {
        ctx->alpha_floor_dev = nullptr;
        return ST_OK;
    }
    PeerMaxima a;
    memset(&a, 0, sizeof(a));
    for (int p = 0; p < npeers; p++) {
        if (!peers[p] || peers[p] == ctx)
            return fail(ctx, ST_API_ERROR, "bad peer");
        if (peers[p]->transformed < 1)
            return fail(ctx, ST_API_ERROR, "a peer's transform stage is not enqueued yet");
        const int st = hydamd_wait_for(ctx, peers[p]);
        if (st != ST_OK)
            return st;
        a.max[p] = peers[p]->alpha_max;
        a.slots[p] = peers[p]->transformed;
    }
    a.count = npeers;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->peer_floor)
        HIP_TRY(ctx, hipMalloc(&ctx->peer_floor, sizeof(uint32_t)));
    hipLaunchKernelGGL(k_floor_from_peers, dim3(1), dim3(64), 0, ctx->stream, a, ctx->peer_floor);
    HIP_TRY(ctx, hipGetLastError());
    ctx->alpha_floor_dev = ctx->peer_floor;
    return ST_OK;
}

/* What the floor kernel read through peer access against what the peers' own devices hold: the host copies every peer's
 * per-LF-group maxima from ITS device (no peer read involved) and the floor ctx's table kernel was given from ctx's, and
 * compares.  Waits for ctx's and the peers' frames.  *ok = 1: the peer read saw what the owners wrote. */
int hydamd_verify_floor(HydAmdContext *ctx, int npeers, HydAmdContext *const *peers, int *ok) {
    if (!ctx || !ok || npeers < 1 || npeers > HYDAMD_MAX_PEERS || !peers)
        return ctx ? fail(ctx, ST_API_ERROR, "bad peer list") : ST_API_ERROR;
    if (!ctx->peer_floor || ctx->alpha_floor_dev != ctx->peer_floor)
        return fail(ctx, ST_API_ERROR, "no floor from peers to verify: hydamd_alphabet_floor_from_peers first");
    uint32_t want = 0;
    for (int p = 0; p < npeers; p++) {
        HydAmdContext *o = peers[p];
        if (!o || o == ctx || o->transformed < 1)
            return fail(ctx, ST_API_ERROR, "bad peer");
        HIP_TRY(o, hipSetDevice(o->device));
        const int st = wait_for_frame(o); /* (the maxima of a frame that outgrew its token arrays are rewritten by its rerun) */
        if (st != ST_OK)
            return st;
        uint32_t m[HYDAMD_MAX_LF_GROUPS];
        const int n = o->transformed < HYDAMD_MAX_LF_GROUPS ? o->transformed : HYDAMD_MAX_LF_GROUPS;
        HIP_TRY(o, hipMemcpy(m, o->alpha_max, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (int i = 0; i < n; i++)
            want = m[i] > want ? m[i] : want;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    uint32_t got = 0;
    HIP_TRY(ctx, hipMemcpy(&got, ctx->peer_floor, sizeof(got), hipMemcpyDeviceToHost));
#ifdef HYD_TEST_HOOKS
    if (const char *e = getenv("HYDAMD_TEST_CORRUPT_PEER_FLOOR")) /* test hook: the floor as a stale peer read would leave it */
        if (*e && *e != '0')
            got ^= 1u;
#endif
    *ok = got == want;
    return ST_OK;
}

/* Run the current frame's stages again from the job descriptors the context still holds (what hydamd_sync does by itself
 * for a frame that outgrew its buffers) — for a caller that changed an input of the closing stage after enqueuing it: the
 * drop-in API's through-the-host fallback gives a shard its alphabet floor as a host value when the peer read of it
 * cannot be trusted.  Waits for the stream first; hydamd_sync afterwards as for any frame. */
int hydamd_replay_frame(HydAmdContext *ctx) {
    if (!ctx)
        return ST_API_ERROR;
    if (ctx->slots_per_frame)
        return fail(ctx, ST_API_ERROR, "a batch of frames is not replayed");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int st = join_lf(ctx, 0);
    if (st == ST_OK)
        st = wait_for_frame(ctx);
    if (st != ST_OK)
        return st;
    if (ctx->lf_stream)
        HIP_TRY(ctx, hipStreamSynchronize(ctx->lf_stream));
    return replay_frame(ctx);
}

/* Can every device of the list read every other one's memory?  hyd_send_tile asks BEFORE it deals a frame out to several
 * devices (a frame that cannot be assembled from peer reads stays on one device).  HYDAMD_TEST_NO_P2P=1 answers "no" for
 * any list of two or more entries — how the tests reach the fallback on a box with one GPU (HYD_TEST_HOOKS flavour of the
 * library only). */
int hydamd_peers_reachable(const int *devices, int n) {
    if (!devices || n < 1)
        return 0;
    if (n == 1)
        return 1;
#ifdef HYD_TEST_HOOKS
    if (const char *e = getenv("HYDAMD_TEST_NO_P2P"))
        if (*e && *e != '0')
            return 0;
#endif
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    for (int i = 0; i < n; i++) {
        if (devices[i] < 0 || devices[i] >= count)
            return 0;
        for (int j = 0; j < n; j++) {
            if (devices[i] == devices[j])
                continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[i], devices[j]) != hipSuccess || !can) {
                (void)hipGetLastError();
                return 0;
            }
        }
    }
    return 1;
}

/* HYDAMD_VERIFY_PEERS: a checksum of everything an exported view of `owner`'s frame names — the view's header and slot
 * records, the packed LF streams, the HF sections — computed by a kernel in `reader`'s stream (reader == owner: on the
 * owning device; else: through peer reads, behind hydamd_wait_for(reader, owner)).  Both sides run the same kernel on the
 * same addresses, so the two sums differ exactly when the reading device saw other bytes than the owner wrote (a stale
 * line of remote memory in its L2, a release that did not reach the fabric). */
namespace {
__device__ __forceinline__ unsigned long long mix_words(const uint8_t *p, unsigned long long bytes, unsigned long long salt) {
    const unsigned long long words = bytes >> 3;
    const unsigned long long *w = (const unsigned long long *)p;
    unsigned long long s = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (unsigned long long)gridDim.x * blockDim.x)
        s += (w[i] ^ salt) * (2ull * i + 1ull);
    if (blockIdx.x == 0 && threadIdx.x < (bytes & 7ull))
        s += ((unsigned long long)p[(words << 3) + threadIdx.x] + 1ull) * (salt | 1ull) * (threadIdx.x + 3ull);
    return s;
}
__global__ __launch_bounds__(256) void k_view_checksum(const uint8_t *view, unsigned long long view_bytes, const uint8_t *lf,
                                                       const unsigned long long *lf_total, const uint8_t *hf,
                                                       const uint64_t *hf_total, unsigned long long *out) {
    unsigned long long s = mix_words(view, view_bytes, 0x9E3779B97F4A7C15ull);
    s += mix_words(lf, *lf_total, 0xC2B2AE3D27D4EB4Full);
    s += mix_words(hf, (unsigned long long)*hf_total, 0x165667B19E3779F9ull);
#pragma unroll
    for (int d = 32; d; d >>= 1)
        s += (unsigned long long)__shfl_xor((long long)s, d);
    if ((threadIdx.x & 63) == 0)
        atomicAdd(out, s);
}
} /* namespace */

int hydamd_verify_enqueue(HydAmdContext *reader, HydAmdContext *owner, int num_slots, int index) {
    if (!reader || !owner || index < 0 || index >= HYDAMD_MAX_PEERS)
        return ST_API_ERROR;
    if (!owner->own_blob || num_slots < 1 || num_slots > owner->max_slots)
        return fail(reader, ST_API_ERROR, "verification needs the owner's frame exported as a view first");
    if (reader != owner) {
        const int st = enable_peer_reads(reader, owner->device);
        if (st != ST_OK)
            return st;
    }
    HIP_TRY(reader, hipSetDevice(reader->device));
    if (!reader->verify_sums)
        HIP_TRY(reader, hipMalloc(&reader->verify_sums, HYDAMD_MAX_PEERS * sizeof(unsigned long long)));
    HIP_TRY(reader, hipMemsetAsync(reader->verify_sums + index, 0, sizeof(unsigned long long), reader->stream));
    const size_t view_bytes = sizeof(HydAmdBlobHeader) + (size_t)num_slots * sizeof(HydAmdBlobSlot);
    hipLaunchKernelGGL(k_view_checksum, dim3(64), dim3(256), 0, reader->stream, (const uint8_t *)owner->own_blob,
                       (unsigned long long)view_bytes, (const uint8_t *)owner->lf_packed, owner->lf_total,
                       (const uint8_t *)owner->payload, owner->total, reader->verify_sums + index);
    HIP_TRY(reader, hipGetLastError());
    return ST_OK;
}

int hydamd_verify_read(HydAmdContext *reader, int index, unsigned long long *sum) {
    if (!reader || !sum || index < 0 || index >= HYDAMD_MAX_PEERS || !reader->verify_sums)
        return ST_API_ERROR;
    HIP_TRY(reader, hipSetDevice(reader->device));
    HIP_TRY(reader, hipStreamSynchronize(reader->stream));
    HIP_TRY(reader, hipMemcpy(sum, reader->verify_sums + index, sizeof(*sum), hipMemcpyDeviceToHost));
#ifdef HYD_TEST_HOOKS
    if (const char *e = getenv("HYDAMD_TEST_CORRUPT_PEER_VIEW")) /* test hook: the sum of view `index` as a faulty link would leave it */
        if (*e && atoi(e) == index)
            *sum ^= 1ull;
#endif
    return ST_OK;
}

size_t hydamd_blob_bound(HydAmdContext *ctx, int num_slots) {
    if (!ctx || num_slots < 1 || num_slots > ctx->max_slots)
        return 0;
    const size_t lf = (size_t)num_slots * HYDK_LF_BITWORDS * sizeof(uint32_t);
    return sizeof(HydAmdBlobHeader) + (size_t)num_slots * sizeof(HydAmdBlobSlot) + lf + 16 + ctx->payload_cap + 16;
}

static int export_frame(HydAmdContext *ctx, int num_slots, void *device_dst, size_t capacity, int view) {
    if (!ctx || !device_dst)
        return ST_API_ERROR;
    if (num_slots < 1 || num_slots > ctx->coded || num_slots != ctx->slots_finished)
        return fail(ctx, ST_API_ERROR, "export needs the entropy stage of the same slots enqueued first");
    if (ctx->slots_per_frame)
        return fail(ctx, ST_API_ERROR, "a batch of frames is not exported as one blob: its sections and LF streams stay in the context's buffers");
    if (ctx->lf_on_device && (ctx->lf_need_gather || ctx->lf_slots != num_slots))
        return fail(ctx, ST_API_ERROR, "export needs the frame's LF streams packed (hydamd_run_entropy does it)");
    if (capacity < sizeof(HydAmdBlobHeader))
        return fail(ctx, ST_API_ERROR, "blob buffer smaller than its header");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hydk::launch_export(ctx->d_jobs, ctx->tables, ctx->group_bits, ctx->lf_streams, ctx->payload, ctx->total,
                                     (const uint8_t *)ctx->lf_packed, ctx->lf_total, ctx->status, num_slots,
                                     ctx->lf_on_device ? 1 : 0, (uint8_t *)device_dst, capacity, view, ctx->stream));
    return ST_OK;
}

int hydamd_export_frame(HydAmdContext *ctx, int num_slots, void *device_dst, size_t capacity) {
    return export_frame(ctx, num_slots, device_dst, capacity, 0);
}

int hydamd_export_frame_owned(HydAmdContext *ctx, int num_slots, const void **blob_dev, size_t *capacity) {
    if (!ctx || !blob_dev || !capacity)
        return ST_API_ERROR;
    if (num_slots < 1 || num_slots > ctx->max_slots)
        return fail(ctx, ST_API_ERROR, "slot count out of range");
    if (!ctx->lf_on_device)
        return fail(ctx, ST_API_ERROR, "the owned blob needs the LF coder on");
    /* a VIEW: header and slot records; the packed LF streams and HF sections stay in the context's own buffers and
     * the header names their addresses — nothing of the frame's bulk is copied */
    const size_t need = sizeof(HydAmdBlobHeader) + (size_t)num_slots * sizeof(HydAmdBlobSlot);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (need > ctx->own_blob_cap) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); /* an assembly of the previous frame may still be reading the old one */
        if (ctx->own_blob)
            (void)hipFree(ctx->own_blob);
        ctx->own_blob = nullptr;
        ctx->own_blob_cap = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->own_blob, need));
        ctx->own_blob_cap = need;
    }
    const int st = export_frame(ctx, num_slots, ctx->own_blob, ctx->own_blob_cap, 1);
    if (st != ST_OK)
        return st;
    *blob_dev = ctx->own_blob;
    *capacity = ctx->own_blob_cap;
    return ST_OK;
}

/* A small frame's results in ONE copy.  The drop-in API's host-assembled frames (tile mode: one LF group per frame) used to
 * read tables, section sizes, LF stream records, LF bytes and HF bytes back one hipMemcpy at a time — six round trips of
 * ~17 us behind a 1.1 ms chain.  hydamd_stage_frame_blob enqueues the export of the frame's self-contained blob into a
 * buffer of the context's; hydamd_read_frame_blob (after hydamd_sync) copies exactly its bytes into pinned host memory. */
int hydamd_stage_frame_blob(HydAmdContext *ctx, int num_slots) {
    if (!ctx)
        return ST_API_ERROR;
    if (num_slots < 1 || num_slots > ctx->max_slots)
        return fail(ctx, ST_API_ERROR, "slot count out of range");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t need = hydamd_blob_bound(ctx, num_slots);
    if (need > ctx->stage_blob_cap) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->stage_blob)
            (void)hipFree(ctx->stage_blob);
        ctx->stage_blob = nullptr;
        ctx->stage_blob_cap = 0;
        HIP_TRY(ctx, hipMalloc(&ctx->stage_blob, need));
        ctx->stage_blob_cap = need;
    }
    ctx->stage_slots = 0;
    const int st = export_frame(ctx, num_slots, ctx->stage_blob, ctx->stage_blob_cap, 0);
    if (st == ST_OK)
        ctx->stage_slots = num_slots;
    return st;
}

int hydamd_read_frame_blob(HydAmdContext *ctx, int num_slots, const void **host_blob, size_t *size) {
    if (!ctx || !host_blob || !size)
        return ST_API_ERROR;
    if (!ctx->results_valid || ctx->stage_slots != num_slots || !ctx->lf_on_device)
        return fail(ctx, ST_API_ERROR, "no staged blob of this frame: hydamd_stage_frame_blob, then hydamd_sync");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t lf_off = sizeof(HydAmdBlobHeader) + (size_t)num_slots * sizeof(HydAmdBlobSlot);
    const size_t hf_off = (lf_off + (size_t)ctx->h_lf_total + 15u) & ~(size_t)15u;
    size_t total = hf_off + (size_t)ctx->h_total;
    if (total > ctx->stage_blob_cap) {
        /* the frame outgrew the context's buffers and was rerun inside hydamd_sync with larger ones: what was staged is the
         * first run's (its header says "incomplete") and the rerun's results do not fit the staging buffer sized before it.
         * No blob: the caller reads the results the separate way */
        *host_blob = nullptr;
        *size = 0;
        return ST_OK;
    }
    for (int attempt = 0; attempt < 2; attempt++) {
        if (total > ctx->stage_host_cap) {
            if (ctx->stage_host)
                (void)hipHostFree(ctx->stage_host);
            ctx->stage_host = nullptr;
            ctx->stage_host_cap = 0;
            const size_t want = total + (total >> 1) + 65536;
            HIP_TRY(ctx, hipHostMalloc((void **)&ctx->stage_host, want, hipHostMallocDefault));
            ctx->stage_host_cap = want;
        }
        HIP_TRY(ctx, hipMemcpy(ctx->stage_host, ctx->stage_blob, total, hipMemcpyDeviceToHost));
        const HydAmdBlobHeader *h = (const HydAmdBlobHeader *)ctx->stage_host;
        if (h->total_bytes <= total || h->total_bytes > ctx->stage_blob_cap)
            break;
        total = (size_t)h->total_bytes; /* (the blob says it is longer than the published totals implied: take it at its word once) */
    }
    *host_blob = ctx->stage_host;
    *size = total;
    return ST_OK;
}

HydAmdAssembler *hydamd_context_assembler(HydAmdContext *ctx) {
    if (!ctx)
        return nullptr;
    if (!ctx->assembler) {
        int st = ST_OK;
        ctx->assembler = hydamd_assembler_create(ctx->device, &st);
        if (!ctx->assembler)
            (void)fail(ctx, st, "frame assembler could not be created");
    }
    return ctx->assembler;
}

/* HYDAMD_WAVE_FORM_EMITS=1: form 4 as until round 4, the chain kernel writing its own bits into the reversed buffers (A/B) */
static bool wave_form_defers() {
    static const bool v = !(getenv("HYDAMD_WAVE_FORM_EMITS") && atoi(getenv("HYDAMD_WAVE_FORM_EMITS")) != 0);
    return v;
}

/* Measurement and test hooks live in the HYD_TEST_HOOKS flavour of the library only (hydrium_amd/lib/libhydrium_probe.so,
 * loaded explicitly by scripts/pipe_probe.py and the tests that need them): the shipped libhydrium.so.0 reads none of
 * these variables and never returns HYD_OK for a frame whose stages disagree.
 *   HYDAMD_DEBUG_SKIP (bit mask; the frame's bytes are then stale or wrong): leave a stage of the closing sequence out of
 *   the stream: 1 table kernel, 2 rANS chains, 4 section scan + emit, 8 the LF coder's kernels, 16 stand-ins in the
 *   chains' place (HYDAMD_DEBUG_STANDIN: sleep | valu | lds | both | valu4; HYDAMD_DEBUG_SLEEP_US / _LDS / _VGPRS / _WGS,
 *   HYDAMD_DEBUG_STANDIN_STEPS).  scripts/pipe_probe.py prices each stage's share of the pipelined frame rate with it. */
#ifdef HYD_TEST_HOOKS
namespace {
__global__ __launch_bounds__(64) void k_sleep_probe(unsigned long long ticks_100mhz) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks_100mhz)
        __builtin_amdgcn_s_sleep(127);
}
/* the same sleeper holding REGS vector registers as well (a chain wavefront holds ~110): which of a chain's holdings — LDS,
 * registers, or what it does — costs the loop its transform workgroups?  (profiles/r05_pipeline_bounds.txt) */
#define HYDK_SLEEP_PROBE_REGS(NAME, REGS)                                                                      \
    __global__ __launch_bounds__(64) void NAME(unsigned long long ticks_100mhz, uint32_t *sink) {               \
        uint32_t r[REGS];                                                                                     \
        _Pragma("unroll") for (int i = 0; i < REGS; i++) r[i] = threadIdx.x * (uint32_t)(i + 1);              \
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();                                       \
        while (__builtin_amdgcn_s_memrealtime() - t0 < ticks_100mhz) {                                        \
            _Pragma("unroll") for (int i = 0; i < REGS; i++) asm volatile("" : "+v"(r[i])); /* stays in a vector register */ \
            __builtin_amdgcn_s_sleep(127);                                                                    \
        }                                                                                                     \
        uint32_t s = 0;                                                                                       \
        _Pragma("unroll") for (int i = 0; i < REGS; i++) s ^= r[i];                                           \
        if (s == 0xFFFFFFFFu && sink)                                                                         \
            *sink = s;                                                                                        \
    }
HYDK_SLEEP_PROBE_REGS(k_sleep_probe_regs104, 104)
HYDK_SLEEP_PROBE_REGS(k_sleep_probe_regs124, 124) /* allocated as 128: three transform wavefronts (3 x 120) still fit beside it */
HYDK_SLEEP_PROBE_REGS(k_sleep_probe_regs160, 160) /* as the lane-form chain with HYDK_LANE_PIPE 2 (164): two fit */
HYDK_SLEEP_PROBE_REGS(k_sleep_probe_regs40, 40)
#undef HYDK_SLEEP_PROBE_REGS

/* Stand-ins that DO one part of what a lane-form chain wavefront does, for `steps` steps, holding 96 registers and the
 * launch's dynamic LDS (round 6, VERDICT r5 task 1: which part of a chain's work costs the loop?).
 *   VALU: per step the chain's 17 vector instructions (same classes, one dependent chain through x), no LDS access;
 *   LDS : per step one ds_read_b128 of a lane-random 16-byte row in the first 32 KB (the operand rows: 64 lanes, random
 *         rows, bank conflicts as in the chain) and one dependent ds_read_u16 (the slot lookup), then s_sleep for the
 *         rest of the step; no arithmetic beyond the address generator;
 *   QUARTER: the VALU stand-in's work at a quarter of the rate (three steps in four are slept away): four such
 *         wavefronts, one per SIMD of a 256-thread workgroup, issue what ONE chain issues — the same instruction load
 *         spread evenly over a compute unit's four SIMDs instead of concentrated on one. */
__global__ __launch_bounds__(256) void k_chain_standin(int mode /* 1 VALU | 2 LDS | 4 QUARTER */, uint32_t steps, uint32_t lds_bytes,
                                                       uint32_t *sink) {
    const bool VALU = mode & 1, LDS = mode & 2, QUARTER = mode & 4; /* launch-uniform: scalar branches */
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    uint32_t hold[96];
#pragma unroll
    for (int i = 0; i < 96; i++)
        hold[i] = threadIdx.x * (uint32_t)(i + 1);
    __builtin_amdgcn_s_setprio(3);
    uint32_t x = 0x130000u + threadIdx.x * 2654435761u, acc = 0;
    const uint32_t row_mask = (lds_bytes >= 32768u ? 32768u : lds_bytes >= 16u ? lds_bytes : 16u) - 16u;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)s_dyn;
    if (LDS) /* something to read */
        for (uint32_t i = threadIdx.x * 4u; i + 4u <= lds_bytes; i += blockDim.x * 4u)
            *(uint32_t *)(s_dyn + i) = i * 2246822519u;
    __syncthreads();
    for (uint32_t s = 0; s < steps; s++) {
        if (QUARTER && (s & 3u) != (threadIdx.x >> 6)) { /* this wavefront's turn is every fourth step */
            __builtin_amdgcn_s_sleep(2);
            continue;
        }
        uint32_t t0, t1, q, a, b, sm, fl = acc, row = 0, slot = 0;
        if (LDS) {
            const uint32_t radr = lds0 + ((x * 2654435761u >> 7) & row_mask & ~15u);
            uint4 o;
            asm volatile("ds_read_b128 %0, %1" : "=v"(o) : "v"(radr));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            row = o.x ^ o.y ^ o.z ^ o.w;
        }
        if (VALU) {
            asm volatile("v_and_or_b32 %[x], %[x], %[k], %[h0]\n\t"
                         "v_mul_hi_u32 %[q], %[x], %[h1]\n\t"
                         "v_mad_i32_i24 %[t0], %[q], %[h2], %[x]\n\t"
                         "v_add_co_u32 %[t1], vcc, %[t0], %[h2]\n\t"
                         "v_min_u32 %[t0], %[t0], %[t1]\n\t"
                         "v_lshl_add_u32 %[t1], %[t0], 1, %[h3]\n\t"
                         "v_or_b32 %[a], %[h0], %[t1]\n\t"
                         "v_perm_b32 %[b], %[a], %[t0], %[h1]\n\t"
                         "v_addc_co_u32 %[q], vcc, 0, %[q], vcc\n\t"
                         "v_lshlrev_b32 %[a], 12, %[q]\n\t"
                         "v_cmp_gt_u32 vcc, %[a], %[h2]\n\t"
                         "v_cndmask_b32_sdwa %[b], %[a], %[a], vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
                         "v_cndmask_b32_e64 %[sm], %[k], 0, vcc\n\t"
                         "v_addc_co_u32 %[fl], vcc, %[fl], %[fl], vcc\n\t"
                         "v_xor_b32 %[x], %[b], %[t1]\n\t"
                         "v_add_u32 %[x], %[x], %[sm]\n\t"
                         "v_or_b32 %[x], 0x10000, %[x]"
                         : [x] "+v"(x), [q] "=&v"(q), [t0] "=&v"(t0), [t1] "=&v"(t1), [a] "=&v"(a), [b] "=&v"(b), [sm] "=&v"(sm), [fl] "+v"(fl)
                         : [k] "v"(hold[0] | 0xfffu), [h0] "v"(hold[1]), [h1] "v"(hold[2] | 0x10001u), [h2] "v"(hold[3] | 1u), [h3] "v"(hold[4])
                         : "vcc");
            acc = fl;
        } else {
            x = x * 1664525u + 1013904223u + row;
        }
        if (LDS) {
            const uint32_t sadr = lds0 + (__umulhi(x, lds_bytes > 2u ? lds_bytes - 2u : 1u) & ~1u);
            asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(slot) : "v"(sadr) : "memory");
            x ^= slot;
        }
        if (!VALU)
            __builtin_amdgcn_s_sleep(1); /* 64 clocks: what the chain spends issuing */
#pragma unroll
        for (int i = 0; i < 96; i++)
            asm volatile("" : "+v"(hold[i])); /* the registers stay allocated (no instruction is emitted) */
    }
    uint32_t r = x ^ acc;
#pragma unroll
    for (int i = 0; i < 96; i++)
        r ^= hold[i];
    if (r == 0xFFFFFFFFu && sink)
        *sink = r;
}
} /* namespace */
static int debug_skip() {
    static const int v = getenv("HYDAMD_DEBUG_SKIP") ? atoi(getenv("HYDAMD_DEBUG_SKIP")) : 0;
    return v;
}
static int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}
/* the stand-ins of HYDAMD_DEBUG_SKIP & 16 for `count` LF groups, in the chains' place in ctx's stream */
static void launch_chain_standins(HydAmdContext *ctx, int count) {
    static const int wgs = env_int("HYDAMD_DEBUG_SLEEP_WGS", 1);
    static const int lds = env_int("HYDAMD_DEBUG_SLEEP_LDS", 0);
    static const unsigned long long ticks = 100ull * (unsigned long long)env_int("HYDAMD_DEBUG_SLEEP_US", 2500);
    static const int regs = env_int("HYDAMD_DEBUG_SLEEP_VGPRS", 0);
    static const int steps = env_int("HYDAMD_DEBUG_STANDIN_STEPS", 30000);
    static const char *kind = getenv("HYDAMD_DEBUG_STANDIN") ? getenv("HYDAMD_DEBUG_STANDIN") : "sleep";
    const dim3 grid(wgs > 0 ? wgs * count : 1);
#define HYDK_STANDIN(MODE, THREADS)                                                                                         \
    do {                                                                                                                    \
        if (lds > 65536)                                                                                                    \
            (void)hipFuncSetAttribute((const void *)k_chain_standin, hipFuncAttributeMaxDynamicSharedMemorySize, lds);      \
        hipLaunchKernelGGL(k_chain_standin, grid, dim3(THREADS), (size_t)lds, ctx->stream, MODE, (uint32_t)steps,           \
                           (uint32_t)lds, (uint32_t *)nullptr);                                                             \
    } while (0)
    if (!strcmp(kind, "valu"))
        HYDK_STANDIN(1, 64);
    else if (!strcmp(kind, "lds"))
        HYDK_STANDIN(2, 64);
    else if (!strcmp(kind, "both"))
        HYDK_STANDIN(3, 64);
    else if (!strcmp(kind, "valu4"))
        HYDK_STANDIN(5, 256);
    else if (regs >= 160) {
        if (lds > 65536)
            (void)hipFuncSetAttribute((const void *)k_sleep_probe_regs160, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(k_sleep_probe_regs160, grid, dim3(64), (size_t)lds, ctx->stream, ticks, (uint32_t *)nullptr);
    } else if (regs >= 124) {
        if (lds > 65536)
            (void)hipFuncSetAttribute((const void *)k_sleep_probe_regs124, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(k_sleep_probe_regs124, grid, dim3(64), (size_t)lds, ctx->stream, ticks, (uint32_t *)nullptr);
    } else if (regs >= 100) {
        if (lds > 65536)
            (void)hipFuncSetAttribute((const void *)k_sleep_probe_regs104, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(k_sleep_probe_regs104, grid, dim3(64), (size_t)lds, ctx->stream, ticks, (uint32_t *)nullptr);
    } else if (regs >= 40) {
        if (lds > 65536)
            (void)hipFuncSetAttribute((const void *)k_sleep_probe_regs40, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(k_sleep_probe_regs40, grid, dim3(64), (size_t)lds, ctx->stream, ticks, (uint32_t *)nullptr);
    } else {
        if (lds > 65536)
            (void)hipFuncSetAttribute((const void *)k_sleep_probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL(k_sleep_probe, grid, dim3(64), (size_t)lds, ctx->stream, ticks);
    }
#undef HYDK_STANDIN
}
#else
static constexpr int debug_skip() { return 0; }
static void launch_chain_standins(HydAmdContext *, int) {}
#endif /* HYD_TEST_HOOKS */

/* K2 + the rANS chains for slots [first, first + count); with_lf_codes: the lane-form launch also builds the
 * LF coder's prefix codes of the same slots (their token kernel must be enqueued already) */
static int entropy_range(HydAmdContext *ctx, int first, int count, bool with_lf_codes) {
    const size_t G = HYDK_GROUPS_PER_LFG, g0 = (size_t)first * G;
    {
        ScopedTimer timer(ctx, HYDAMD_K_TABLES);
        /* the LF code construction's passengers ride in the chain kernel's launch (2.5 ms long anyway).  HYDAMD_LF_CODES_RIDE=tables
         * puts them into this launch instead, where they do not ask for the chain kernel's 80 KB of LDS: measured equal in the
         * pipelined loop (143.5 against 143.3 Gpixel/s) and 0.13 ms worse for one frame alone (the table kernel then lasts
         * 0.20 instead of 0.07 ms in front of the chains), so it stays an A/B switch */
        if (!(debug_skip() & 1)) {
        static const bool ride_with_tables = getenv("HYDAMD_LF_CODES_RIDE") && !strcmp(getenv("HYDAMD_LF_CODES_RIDE"), "tables");
        const bool here = with_lf_codes && ride_with_tables;
        HIP_TRY(ctx, hydk::launch_tables(ctx->hist, ctx->tables, ctx->alpha_max, ctx->nclusters, first, count,
                                         ctx->alpha_floor, ctx->alpha_floor_dev,
                                         here ? ctx->lf_hist + (size_t)first * HYDK_LF_CODES : nullptr, ctx->lf_streams + first,
                                         ctx->lf_work + (size_t)first * hydk::lf_work_bytes(), ctx->slots_per_frame, ctx->stream));
        if (here)
            with_lf_codes = false;
        }
    }
    {
        ScopedTimer timer(ctx, HYDAMD_K_RANS);
        const HydkLfJob *jobs = ctx->d_jobs + first;
        bool any_float = false; /* 8-byte records: only the wave form reads them */
        for (int i = first; i < first + count; i++)
            any_float = any_float || ctx->h_jobs[i].fmt == HYDK_FMT_F32;
        const bool lanes = ctx->rans_lanes && !any_float;
        bool emits = lanes; /* the slots' bits are written by k_rans_emit (else: by the chain kernel itself + k_pack_sections) */
        if (with_lf_codes && !lanes)
            return fail(ctx, ST_INTERNAL_ERROR, "LF code construction can only ride with the lane-form entropy stage");
        (void)lanes;
        if (debug_skip() & 16) { /* measurement builds only: stand-ins in the chains' place */
            launch_chain_standins(ctx, count);
        } else if (debug_skip() & 2) {
        } else if (lanes) {
            HIP_TRY(ctx, hydk::launch_rans_lanes(jobs, ctx->sym_count + g0, ctx->tables + first,
                                                 ctx->rans_aux + g0 * ctx->tok_cap, ctx->rans_flags + g0 * (ctx->tok_cap / 16),
                                                 ctx->tok_cap, ctx->rans_final + g0, ctx->group_bits + g0, ctx->preset_bits,
                                                 ctx->nclusters, count, ctx->status,
                                                 with_lf_codes ? ctx->lf_hist + (size_t)first * HYDK_LF_CODES : nullptr,
                                                 ctx->lf_streams + first, ctx->lf_work + (size_t)first * hydk::lf_work_bytes(),
                                                 ctx->stream));
        } else if (!any_float && wave_form_defers()) {
            /* wave per group, bits written by k_rans_emit as for the lane form (round 4: the walk no longer stops after
             * every 64 symbols to scan, pack and store their bits itself) */
            HIP_TRY(ctx, hydk::launch_rans_deferred(jobs, ctx->sym_count + g0, ctx->tables + first, ctx->rans_aux + g0 * ctx->tok_cap,
                                                    ctx->rans_flags + g0 * (ctx->tok_cap / 16), ctx->tok_cap, ctx->rans_final + g0,
                                                    ctx->group_bits + g0, ctx->preset_bits, count, ctx->status, ctx->stream));
            emits = true;
        } else {
            const int st = ensure_bitbuf(ctx);
            if (st != ST_OK)
                return st;
            HIP_TRY(ctx, hydk::launch_rans(jobs, ctx->sym_count + g0, ctx->tables + first,
                                           ctx->bitbuf + g0 * ctx->bit_pitch_words, ctx->bit_pitch_words,
                                           ctx->group_bits + g0, ctx->preset_bits, count, ctx->status, ctx->stream));
        }
        for (int i = first; i < first + count; i++)
            ctx->slot_lanes[i] = emits;
    }
    return ST_OK;
}

int hydamd_run_entropy(HydAmdContext *ctx, int num_slots) {
    if (!ctx)
        return ST_API_ERROR;
    if (num_slots < 1 || num_slots > ctx->max_slots)
        return fail(ctx, ST_API_ERROR, "slot count out of range");
    if (num_slots > ctx->transformed)
        return fail(ctx, ST_API_ERROR, "the entropy stage needs the transform stage of the same slots first");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int count = num_slots * HYDK_GROUPS_PER_LFG;
    ctx->results_valid = false;
    ctx->want_entropy = num_slots;
    /* In-stream LF coder + lane-form entropy stage: the LF coder's 200 us of serial code construction per LF
     * group rides in the chain kernel's launch instead of sitting in the stream on its own — tokens before
     * the entropy stage, offsets + pack behind it. */
    int lf_first = -1, lf_count = 0;
    if (num_slots > ctx->coded && ctx->lf_on_device == 2 && ctx->rans_lanes && ctx->lf_coded == ctx->coded && !ctx->lf_pending &&
        !(debug_skip() & 8)) {
        bool any_float = false;
        for (int i = ctx->coded; i < num_slots; i++)
            any_float = any_float || ctx->h_jobs[i].fmt == HYDK_FMT_F32;
        if (!any_float) {
            lf_first = ctx->coded;
            lf_count = num_slots - ctx->coded;
            ScopedTimer timer(ctx, HYDAMD_K_LF);
            if (!(lf_drop() & 1))
            HIP_TRY(ctx, hydk::launch_lf_front(ctx->d_jobs + lf_first, ctx->lf_recs + (size_t)lf_first * HYDK_LF_SYMBOLS,
                                               ctx->lf_hist + (size_t)lf_first * HYDK_LF_CODES,
                                               ctx->lf_work + (size_t)lf_first * hydk::lf_work_bytes(), lf_count, ctx->stream));
        }
    }
    if (num_slots > ctx->coded) {
        const int st = entropy_range(ctx, ctx->coded, num_slots - ctx->coded, lf_count > 0 && !(lf_drop() & 2));
        if (st != ST_OK)
            return st;
        ctx->coded = num_slots;
    }
    if (!(debug_skip() & 4)) {
        /* section sizes -> byte offsets, then the sections themselves: lane-form slots write their bits
         * straight into place, wave-form slots copy theirs out of the reversed bit buffers */
        ScopedTimer timer(ctx, HYDAMD_K_PACK);
        HIP_TRY(ctx, hydk::launch_scan(ctx->group_bits, count, ctx->offsets, ctx->total, ctx->payload, ctx->payload_cap, 1,
                                       ctx->status, ctx->stream));
        for (int first = 0; first < num_slots;) {
            int last = first;
            while (last + 1 < num_slots && ctx->slot_lanes[last + 1] == ctx->slot_lanes[first])
                last++;
            const int n = last - first + 1;
            const size_t g0 = (size_t)first * HYDK_GROUPS_PER_LFG;
            if (ctx->slot_lanes[first])
                HIP_TRY(ctx, hydk::launch_rans_emit(ctx->d_jobs + first, ctx->sym_count + g0, ctx->rans_aux + g0 * ctx->tok_cap,
                                                    ctx->rans_flags + g0 * (ctx->tok_cap / 16), ctx->tok_cap,
                                                    ctx->rans_final + g0, ctx->group_bits + g0, ctx->offsets + g0, ctx->payload,
                                                    ctx->preset_bits, n, ctx->status, ctx->stream));
            else
                HIP_TRY(ctx, hydk::launch_pack(ctx->bitbuf + g0 * ctx->bit_pitch_words, ctx->bit_pitch_words,
                                               ctx->group_bits + g0, ctx->offsets + g0, ctx->payload,
                                               n * HYDK_GROUPS_PER_LFG, ctx->status, ctx->stream));
            first = last + 1;
        }
    }
    if (lf_count > 0) {
        ScopedTimer timer(ctx, HYDAMD_K_LF);
        if (!(lf_drop() & 4))
        HIP_TRY(ctx, hydk::launch_lf_back(ctx->d_jobs + lf_first, ctx->lf_recs + (size_t)lf_first * HYDK_LF_SYMBOLS,
                                          ctx->lf_streams + lf_first, ctx->lf_bits + (size_t)lf_first * HYDK_LF_BITWORDS,
                                          ctx->lf_work + (size_t)lf_first * hydk::lf_work_bytes(), lf_count, ctx->stream));
        ctx->lf_coded = num_slots;
        ctx->lf_need_gather = true;
    }
    {
        const int st = join_lf(ctx, num_slots, true);
        if (st != ST_OK)
            return st;
    }
    /* one single-wave kernel writes the frame's totals and status straight into pinned host memory */
    {
        uint64_t px = 0;
        for (int i = 0; i < num_slots; i++)
            px += (uint64_t)ctx->h_jobs[i].width * (uint64_t)ctx->h_jobs[i].height;
        ctx->published_pixels = px;
    }
    HIP_TRY(ctx, hydk::launch_publish(ctx->total, ctx->h_total_pinned, ctx->lf_total_unpublished ? ctx->lf_total : nullptr,
                                      ctx->h_lf_total_pinned, ctx->status, ctx->h_status_pinned, ctx->stream));
#ifdef HYD_TEST_HOOKS
    {
        /* HYDAMD_DEBUG_EXTRA_LAUNCHES=n: n more single-wavefront kernels that do nothing, at the end of the closing sequence —
         * what does a kernel BOUNDARY cost a stream inside the pipelined loop, whatever the kernel does? */
        static const int extra = env_int("HYDAMD_DEBUG_EXTRA_LAUNCHES", 0);
        for (int i = 0; i < extra; i++)
            HIP_TRY(ctx, hydk::launch_publish(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, ctx->stream));
    }
#endif
    if (ctx->lf_total_unpublished) /* hydamd_sync_lf waits for this event and then reads the LF total */
        HIP_TRY(ctx, hipEventRecord(ctx->lf_ready, ctx->stream));
    ctx->lf_total_unpublished = false;
    ctx->status_published = true;
    ctx->slots_finished = num_slots;
    return ST_OK;
}

int hydamd_submit_lf_group(HydAmdContext *ctx, int slot) {
    int st = check_slot(ctx, slot);
    if (st != ST_OK)
        return st;
    if (slot != ctx->transformed)
        return fail(ctx, ST_API_ERROR, "LF groups are submitted early in slot order, one at a time");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->results_valid = false;
    /* Only the transform kernel runs early.  The rANS stage is latency-bound by its longest chain
     * (2 ms whether it codes one LF group or sixteen) and the LF coder by its one workgroup per LF
     * group (0.6 ms likewise): launched per LF group they would queue up sixteen deep behind the
     * uploads (measured: 53 ms instead of 20 ms per 8K frame); they stay batched launches at
     * hydamd_finish_frame, where the LF coder hides behind the rANS stage. */
    st = transform_range(ctx, slot, 1);
    if (st == ST_OK)
        ctx->transformed = slot + 1;
    return st;
}

int hydamd_run_lf_coder(HydAmdContext *ctx, int num_slots, int last) {
    if (!ctx)
        return ST_API_ERROR;
    if (!ctx->lf_on_device)
        return fail(ctx, ST_API_ERROR, "the LF coder is off");
    if (num_slots < 1 || num_slots > ctx->max_slots)
        return fail(ctx, ST_API_ERROR, "slot count out of range");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->results_valid = false;
    if (num_slots > ctx->transformed) { /* the LF coder reads the LF ints the transform stage writes */
        const int st = transform_range(ctx, ctx->transformed, num_slots - ctx->transformed);
        if (st != ST_OK)
            return st;
        ctx->transformed = num_slots;
    }
    if (ctx->lf_pending) {
        /* earlier LF groups of this frame were coded on the side stream (a replayed frame): everything
         * from here on, the packing of the LF streams included, runs in the main stream behind them */
        HIP_TRY(ctx, hipEventRecord(ctx->lf_join, ctx->lf_stream));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->lf_join, 0));
        ctx->lf_pending = false;
    }
    if (num_slots > ctx->lf_coded) {
        const int st = lf_range(ctx, ctx->lf_coded, num_slots - ctx->lf_coded, false);
        if (st != ST_OK)
            return st;
        ctx->lf_coded = num_slots;
    }
    return last ? join_lf(ctx, num_slots) : ST_OK;
}

int hydamd_sync_lf(HydAmdContext *ctx) {
    if (!ctx)
        return ST_API_ERROR;
    if (!ctx->lf_on_device || ctx->lf_need_gather || ctx->lf_slots == 0)
        return fail(ctx, ST_API_ERROR, "no finished LF coder run to wait for");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipEventSynchronize(ctx->lf_ready));
    ctx->h_lf_total = *ctx->h_lf_total_pinned;
    ctx->lf_results_valid = true;
    return ST_OK;
}

int hydamd_finish_frame(HydAmdContext *ctx, int num_slots) {
    int st = hydamd_run_transform(ctx, num_slots);
    if (st != ST_OK)
        return st;
    return hydamd_run_entropy(ctx, num_slots);
}

int hydamd_sync(HydAmdContext *ctx) {
    if (!ctx)
        return ST_API_ERROR;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    {
        const int st = join_lf(ctx, 0); /* transform stage without an entropy stage: just rejoin the side stream */
        if (st != ST_OK)
            return st;
    }
    {
        const int st = wait_for_frame(ctx);
        if (st != ST_OK)
            return st;
    }
    drain_timers(ctx);
    ctx->h_total = *ctx->h_total_pinned;
    if (ctx->slots_finished > 0 && ctx->published_pixels) { /* this frame's (or batch's) pair, for the next frame's curve-gather choice */
        ctx->seen_bytes = ctx->h_total;
        ctx->seen_pixels = ctx->published_pixels;
    }
    ctx->h_lf_total = *ctx->h_lf_total_pinned;
    ctx->lf_results_valid = ctx->lf_on_device && !ctx->lf_need_gather && ctx->lf_slots > 0;
    ctx->h_status = *ctx->h_status_pinned;
    if (ctx->h_status & HYDK_STATUS_BAD_SAMPLE)
        return fail(ctx, ST_API_ERROR, "Invalid NaN Float");
    /* (the measurement flavour's stage skipping makes the stages disagree by design; the shipped library has no such switch) */
    if ((ctx->h_status & HYDK_STATUS_INCONSISTENT) && !debug_skip())
        return fail(ctx, ST_INTERNAL_ERROR, "a section holds more bits than its rANS chain counted (the device stages disagree about a group)");
    ctx->results_valid = true;
    return ST_OK;
}

size_t hydamd_payload_size(HydAmdContext *ctx) { return ctx && ctx->results_valid ? (size_t)ctx->h_total : 0; }

size_t hydamd_payload_capacity(HydAmdContext *ctx) { return ctx ? ctx->payload_cap : 0; }

unsigned hydamd_token_capacity(HydAmdContext *ctx) { return ctx ? ctx->tok_cap : 0; }

unsigned hydamd_overflow_reruns(HydAmdContext *ctx) { return ctx ? ctx->overflow_reruns : 0; }

unsigned hydamd_grown_ahead(HydAmdContext *ctx) { return ctx ? ctx->grown_ahead : 0; }

const uint8_t *hydamd_payload_device(HydAmdContext *ctx) { return ctx ? ctx->payload : nullptr; }

int hydamd_read_payload(HydAmdContext *ctx, uint8_t *dst, size_t capacity) {
    if (!ctx || !ctx->results_valid)
        return fail(ctx, ST_API_ERROR, "results not ready: call hydamd_finish_frame and hydamd_sync first");
    if (capacity < ctx->h_total)
        return fail(ctx, ST_API_ERROR, "payload buffer too small");
    if (ctx->h_total)
        HIP_TRY(ctx, hipMemcpy(dst, ctx->payload, ctx->h_total, hipMemcpyDeviceToHost));
    return ST_OK;
}

int hydamd_read_sections(HydAmdContext *ctx, int slot, uint32_t bits[HYDAMD_GROUPS_PER_LFG],
                         uint64_t offsets[HYDAMD_GROUPS_PER_LFG]) {
    int st = check_slot(ctx, slot);
    if (st != ST_OK)
        return st;
    HIP_TRY(ctx, hipMemcpy(bits, ctx->group_bits + (size_t)slot * HYDK_GROUPS_PER_LFG,
                           HYDK_GROUPS_PER_LFG * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (offsets)
        HIP_TRY(ctx, hipMemcpy(offsets, ctx->offsets + (size_t)slot * HYDK_GROUPS_PER_LFG,
                               HYDK_GROUPS_PER_LFG * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return ST_OK;
}

int hydamd_read_tables(HydAmdContext *ctx, int slot, uint32_t freq[HYDAMD_MAX_CLUSTERS][HYDAMD_ALPHABET],
                       uint32_t alphabet[HYDAMD_MAX_CLUSTERS], uint32_t *log_alphabet_size,
                       uint32_t *running_max_alphabet) {
    int st = check_slot(ctx, slot);
    if (st != ST_OK)
        return st;
    const HydkTables *t = ctx->tables + slot;
    struct {
        uint32_t alphabet[HYDK_MAX_CLUSTERS];
        uint32_t log_alphabet_size, running_max_alphabet, error, pad;
    } tail;
    HIP_TRY(ctx, hipMemcpy(freq, t->freq, sizeof(t->freq), hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(&tail, t->alphabet, sizeof(tail), hipMemcpyDeviceToHost));
    memcpy(alphabet, tail.alphabet, sizeof(tail.alphabet));
    if (log_alphabet_size)
        *log_alphabet_size = tail.log_alphabet_size;
    if (running_max_alphabet)
        *running_max_alphabet = tail.running_max_alphabet;
    if (tail.error)
        return fail(ctx, ST_INTERNAL_ERROR, "ANS table construction failed on the device");
    return ST_OK;
}

int hydamd_read_dc(HydAmdContext *ctx, int slot, int32_t *dst, size_t vbw, size_t vbh) {
    int st = check_slot(ctx, slot);
    if (st != ST_OK)
        return st;
    if (vbw > HYDK_DC_PITCH || vbh > HYDK_DC_PITCH)
        return fail(ctx, ST_API_ERROR, "DC plane larger than an LF group");
    const int32_t *src = ctx->dc + (size_t)slot * 3 * HYDK_DC_PITCH * HYDK_DC_PITCH;
    for (int c = 0; c < 3; c++)
        HIP_TRY(ctx, hipMemcpy2D(dst + (size_t)c * vbw * vbh, vbw * sizeof(int32_t),
                                 src + (size_t)c * HYDK_DC_PITCH * HYDK_DC_PITCH, HYDK_DC_PITCH * sizeof(int32_t),
                                 vbw * sizeof(int32_t), vbh, hipMemcpyDeviceToHost));
    return ST_OK;
}

int hydamd_set_lf_coder(HydAmdContext *ctx, int on_device) {
    if (!ctx)
        return ST_API_ERROR;
    ctx->lf_on_device = on_device == 2 ? 2 : on_device != 0;
    return ST_OK;
}

int hydamd_lf_coder(HydAmdContext *ctx) { return ctx ? ctx->lf_on_device : 0; }

int hydamd_read_lf_stream(HydAmdContext *ctx, int slot, uint8_t lengths[HYDAMD_LF_CODES], uint32_t *alphabet,
                          uint32_t *run_pairs, uint32_t *bit_count) {
    int st = check_slot(ctx, slot);
    if (st != ST_OK)
        return st;
    if (!ctx->lf_on_device || !ctx->lf_results_valid)
        return fail(ctx, ST_API_ERROR, "no device-coded LF stream: the LF coder is off or the frame was not synchronised");
    HydkLfStream h;
    HIP_TRY(ctx, hipMemcpy(&h, ctx->lf_streams + slot, sizeof(h), hipMemcpyDeviceToHost));
    if (h.error)
        return fail(ctx, ST_INTERNAL_ERROR, "LF code construction failed on the device");
    memcpy(lengths, h.lengths, HYDK_LF_CODES);
    if (alphabet)
        *alphabet = h.alphabet;
    if (run_pairs)
        *run_pairs = h.run_pairs;
    if (bit_count)
        *bit_count = h.bit_count;
    return ST_OK;
}

int hydamd_read_lf_bits(HydAmdContext *ctx, int slot, uint8_t *dst, size_t capacity) {
    int st = check_slot(ctx, slot);
    if (st != ST_OK)
        return st;
    if (!ctx->lf_on_device || !ctx->lf_results_valid)
        return fail(ctx, ST_API_ERROR, "no device-coded LF stream: the LF coder is off or the frame was not synchronised");
    if (capacity > (size_t)HYDK_LF_BITWORDS * sizeof(uint32_t))
        return fail(ctx, ST_API_ERROR, "LF bit request too large");
    if (capacity)
        HIP_TRY(ctx, hipMemcpy(dst, ctx->lf_bits + (size_t)slot * HYDK_LF_BITWORDS, capacity, hipMemcpyDeviceToHost));
    return ST_OK;
}

size_t hydamd_lf_payload_size(HydAmdContext *ctx) {
    return ctx && ctx->lf_results_valid && ctx->lf_on_device ? (size_t)ctx->h_lf_total : 0;
}

const uint8_t *hydamd_lf_payload_device(HydAmdContext *ctx) { return ctx ? (const uint8_t *)ctx->lf_packed : nullptr; }

int hydamd_read_lf_payload(HydAmdContext *ctx, uint8_t *dst, size_t capacity) {
    if (!ctx || !ctx->lf_on_device || !ctx->lf_results_valid)
        return fail(ctx, ST_API_ERROR, "no device-coded LF stream: the LF coder is off or the frame was not synchronised");
    if (capacity < ctx->h_lf_total)
        return fail(ctx, ST_API_ERROR, "LF payload buffer too small");
    if (ctx->h_lf_total)
        HIP_TRY(ctx, hipMemcpy(dst, ctx->lf_packed, ctx->h_lf_total, hipMemcpyDeviceToHost));
    return ST_OK;
}

int hydamd_read_lf_streams(HydAmdContext *ctx, int first_slot, int count, HydAmdLfInfo *dst) {
    if (!ctx || !ctx->lf_on_device || !ctx->lf_results_valid)
        return fail(ctx, ST_API_ERROR, "no device-coded LF stream: the LF coder is off or the frame was not synchronised");
    if (first_slot < 0 || count < 1 || first_slot + count > ctx->lf_slots)
        return fail(ctx, ST_API_ERROR, "LF-group slot range out of bounds");
    static_assert(sizeof(HydAmdLfInfo) == sizeof(HydkLfStream), "public and kernel-side LF stream records must agree");
    HIP_TRY(ctx, hipMemcpy(dst, ctx->lf_streams + first_slot, (size_t)count * sizeof(HydkLfStream), hipMemcpyDeviceToHost));
    for (int i = 0; i < count; i++)
        if (dst[i].error)
            return fail(ctx, ST_INTERNAL_ERROR, "LF code construction failed on the device");
    return ST_OK;
}

int hydamd_debug_transform_footprint(HydAmdContext *ctx, int sample_fmt, int *lds_bytes, int *registers) {
    if (!ctx || !lds_bytes || !registers || sample_fmt < 0 || sample_fmt > 2)
        return ST_API_ERROR;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hydk::transform_footprint(sample_fmt, ctx->use_luts, lds_bytes, registers));
    return ST_OK;
}

/* The shader clock right now, in MHz, seen from inside a kernel: a single wavefront spins for ~20 us and compares the shader
 * clock counter (s_memtime) with the constant 100 MHz reference (s_memrealtime).  On a stream of its own, so that it can be
 * asked while the context's frames are running (scripts/pipe_probe.py --clock: the pipelined loop's rate settles lower
 * after its first 150 ms and the clock does not — DESIGN.md 4). */
namespace {
__global__ __launch_bounds__(64) void k_clock_probe(unsigned long long *out) {
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < 2000ull) /* 20 us of the 100 MHz counter */
        r1 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
        out[0] = c1 - c0;
        out[1] = r1 - r0;
    }
}
} /* namespace */

int hydamd_debug_shader_clock_mhz(HydAmdContext *ctx, double *mhz) {
    if (!ctx || !mhz)
        return ST_API_ERROR;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    /* on the context's own device and stream-independent: a side stream and a pinned pair per device, made under a lock
     * (one process may drive several devices from several threads) */
    static std::mutex lock;
    static hipStream_t sides[HYDAMD_MAX_PEERS * 2];
    static unsigned long long *pinned[HYDAMD_MAX_PEERS * 2];
    if (ctx->device < 0 || ctx->device >= HYDAMD_MAX_PEERS * 2)
        return fail(ctx, ST_API_ERROR, "device index out of range");
    std::lock_guard<std::mutex> hold(lock);
    hipStream_t &side = sides[ctx->device];
    unsigned long long *&h = pinned[ctx->device];
    if (!side)
        HIP_TRY(ctx, hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    if (!h)
        HIP_TRY(ctx, hipHostMalloc((void **)&h, 2 * sizeof(unsigned long long), hipHostMallocDefault));
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, side, h);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(side));
    *mhz = h[1] ? 100.0 * (double)h[0] / (double)h[1] : 0.0;
    return ST_OK;
}

int hydamd_debug_lf_code(HydAmdContext *ctx, const uint32_t hist[HYDAMD_LF_CODES], uint8_t lengths[HYDAMD_LF_CODES],
                         uint32_t codes[HYDAMD_LF_CODES], uint32_t *alphabet, uint32_t *error) {
    if (!ctx)
        return ST_API_ERROR;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t spare = (size_t)ctx->max_slots; /* the scratch entry behind the frame's slots */
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    {
        const int st = ensure_lf_stream(ctx);
        if (st != ST_OK)
            return st;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->lf_stream));
    uint32_t *d_hist = ctx->lf_hist + spare * HYDK_LF_CODES;
    HIP_TRY(ctx, hipMemcpy(d_hist, hist, HYDK_LF_CODES * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hydk::launch_lf_huffman_only(d_hist, ctx->lf_streams + spare, ctx->lf_codes, ctx->lf_stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->lf_stream));
    HydkLfStream h;
    HIP_TRY(ctx, hipMemcpy(&h, ctx->lf_streams + spare, sizeof(h), hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(codes, ctx->lf_codes, HYDK_LF_CODES * sizeof(uint32_t), hipMemcpyDeviceToHost));
    memcpy(lengths, h.lengths, HYDK_LF_CODES);
    if (alphabet)
        *alphabet = h.alphabet;
    if (error)
        *error = h.error;
    return ST_OK;
}

int hydamd_read_symbol_counts(HydAmdContext *ctx, int slot, uint32_t counts[HYDAMD_GROUPS_PER_LFG]) {
    int st = check_slot(ctx, slot);
    if (st != ST_OK)
        return st;
    HIP_TRY(ctx, hipMemcpy(counts, ctx->sym_count + (size_t)slot * HYDK_GROUPS_PER_LFG,
                           HYDK_GROUPS_PER_LFG * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return ST_OK;
}

int hydamd_read_tokens(HydAmdContext *ctx, int slot, int group, uint64_t *dst, size_t capacity) {
    int st = check_slot(ctx, slot);
    if (st != ST_OK)
        return st;
    if (group < 0 || group >= HYDK_GROUPS_PER_LFG || capacity > ctx->tok_cap)
        return fail(ctx, ST_API_ERROR, "group or capacity out of range");
    const char *src = ctx->tokens + ((size_t)slot * HYDK_GROUPS_PER_LFG + group) * ctx->tok_cap * ctx->rec_bytes;
    if (ctx->h_jobs[slot].fmt == HYDK_FMT_F32) {
        HIP_TRY(ctx, hipMemcpy(dst, src, capacity * sizeof(uint64_t), hipMemcpyDeviceToHost));
        return ST_OK;
    }
    /* integer input: 4-byte records on the device, widened to the documented form here */
    std::vector<uint32_t> narrow(capacity);
    HIP_TRY(ctx, hipMemcpy(narrow.data(), src, capacity * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < capacity; i++)
        dst[i] = ((uint64_t)(narrow[i] >> 16) << 32) | HYDK_REC32_TO_LO(narrow[i]);
    return ST_OK;
}

int hydamd_read_debug_plane(HydAmdContext *ctx, int which, void *dst, size_t pitch, size_t rows) {
    if (!ctx)
        return ST_API_ERROR;
    const void *src = which == 0 ? (const void *)ctx->dbg_xyb : which == 1 ? (const void *)ctx->dbg_dct
                                                                           : (const void *)ctx->dbg_quant;
    if (!src)
        return fail(ctx, ST_API_ERROR, "context was created without debug planes");
    if (pitch > 2048 || rows > 2048)
        return fail(ctx, ST_API_ERROR, "debug plane request too large");
    for (int c = 0; c < 3; c++)
        HIP_TRY(ctx, hipMemcpy2D((char *)dst + (size_t)c * pitch * rows * 4, pitch * 4,
                                 (const char *)src + (size_t)c * kDbgPlane * 4, 2048 * 4, pitch * 4, rows,
                                 hipMemcpyDeviceToHost));
    return ST_OK;
}

int hydamd_profile(HydAmdContext *ctx, int enable) {
    if (!ctx)
        return ST_API_ERROR;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    drain_timers(ctx);
    ctx->profiling = enable != 0;
    return ST_OK;
}

int hydamd_profile_read(HydAmdContext *ctx, double ms[HYDAMD_K_COUNT], uint64_t launches[HYDAMD_K_COUNT]) {
    if (!ctx)
        return ST_API_ERROR;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    drain_timers(ctx);
    for (int i = 0; i < HYDAMD_K_COUNT; i++) {
        ms[i] = ctx->prof_ms[i];
        launches[i] = ctx->prof_n[i];
        ctx->prof_ms[i] = 0;
        ctx->prof_n[i] = 0;
    }
    return ST_OK;
}

} /* extern "C" */
