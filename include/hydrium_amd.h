/*
 * hydrium_amd.h — additive C-ABI of the MI355X build of libhydrium (not present in the reference).
 *
 * The drop-in boundary is include/libhydrium/libhydrium.h (the reference's nine hyd_* functions).
 * The functions below expose the layer underneath it — the HIP hot path of one LF group at a
 * time — so that callers which already hold pixels in HBM (bench.py, multi-GPU sharding, a video
 * pipeline) can skip the host-pointer API the reference's hyd_send_tile imposes
 * (reference src/include/libhydrium/libhydrium.h:260-262, SURVEY.md §8b "additive extension").
 *
 * One HydAmdContext drives one GPU through one HIP stream.  All calls are asynchronous with
 * respect to the GPU unless stated; hydamd_sync() is the only blocking point.  Plain pointers and
 * sizes only: no C++ or torch types cross this boundary.
 *
 * What each call replaces in the reference (file:line relative to /root/reference/src/libhydrium/):
 *   hydamd_encode_lf_group*   hyd_populate_xyb_buffer (format.c:142), forward_dct (encoder.c:631),
 *                             the HF quantiser (encoder.c:783-823), the LF ints of write_lf_group
 *                             (encoder.c:573,582), initialize_hf_coeffs (encoder.c:689),
 *                             hyd_ans_prepare_frequencies (entropy.c:943) and the per-group
 *                             hyd_ans_write_stream_symbols loop (encoder.c:940-950, entropy.c:1064)
 *   hydamd_finish_frame       the byte-padding + concatenation of encoder.c:973-981
 *   (LF coder, on by default) the LF-coefficient stream of write_lf_group (encoder.c:560-596) with its
 *                             LZ77-as-RLE symbol buffering (entropy.c:473-524), Huffman code construction
 *                             (entropy.c:577-707) and symbol write-out (entropy.c:1003-1021)
 */
#ifndef HYDRIUM_AMD_H_
#define HYDRIUM_AMD_H_

#include <stddef.h>
#include <stdint.h>

#include "libhydrium/libhydrium.h"

#if defined(__GNUC__) || defined(__clang__)
#define HYDAMD_EXPORT __attribute__((visibility("default")))
#else
#define HYDAMD_EXPORT
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define HYDAMD_MAX_CLUSTERS 9      /* clusters owned by one preset */
#define HYDAMD_ALPHABET 128        /* row pitch of the frequency tables */
#define HYDAMD_GROUPS_PER_LFG 64   /* group slots per LF group (8 x 8) */
#define HYDAMD_MAX_LF_GROUPS 255   /* the reference cannot code 256 presets (entropy.c:99) */

typedef struct HydAmdContext HydAmdContext;

#define HYDAMD_LF_CODES 384        /* compact token space of the LF-coefficient stream: [0,256) literals, [256,384) token 16384 + (i - 256) */

/* Kernel classes timed by the optional profiler. */
enum { HYDAMD_K_TRANSFORM = 0, HYDAMD_K_TABLES = 1, HYDAMD_K_RANS = 2, HYDAMD_K_PACK = 3, HYDAMD_K_LF = 4, HYDAMD_K_COUNT = 5 };

/* Number of usable HIP devices (0 when there is none; never fails). */
HYDAMD_EXPORT int hydamd_device_count(void);

/* memcpy on the library's staging threads (HYDAMD_STAGE_THREADS) for moves of several megabytes; plain memcpy below 4 MB */
HYDAMD_EXPORT void hydamd_host_copy(void *dst, const void *src, size_t n);

/*
 * Create a context on `device` with room for `max_lf_groups` LF groups in flight (one frame's
 * worth: every LF group of a one-frame image, or 1 for tile mode).  `linear_light` selects the
 * input transfer LUTs (HYDImageMetadata.linear_light).  `debug_planes` != 0 additionally
 * allocates the XYB / DCT / quantised dump planes used by the parity tests.
 * Returns NULL on failure; *status (if non-NULL) receives a HYDStatusCode-compatible value and
 * hydamd_error(NULL) a description.
 */
HYDAMD_EXPORT HydAmdContext *hydamd_create(int device, int max_lf_groups, int linear_light, int debug_planes, int *status);
HYDAMD_EXPORT void hydamd_destroy(HydAmdContext *ctx);
HYDAMD_EXPORT const char *hydamd_error(HydAmdContext *ctx);

/* Run on a caller-owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); NULL restores the context's own. */
HYDAMD_EXPORT int hydamd_set_stream(HydAmdContext *ctx, void *hip_stream);
HYDAMD_EXPORT void *hydamd_get_stream(HydAmdContext *ctx);

/* 1 if K1 evaluates the format.c LUTs in registers (verified bit-exact at creation), 0 if it gathers from them. */
HYDAMD_EXPORT int hydamd_uses_register_luts(HydAmdContext *ctx);
/* Force the LUT-gather (1) or register (0) variant; for A/B measurements. */
HYDAMD_EXPORT int hydamd_force_luts(HydAmdContext *ctx, int use_luts);
/* XYB evaluation mode of the integer pixel path: 0 registers with v_rcp + one fused Newton step,
 * 1 registers with IEEE division, 2 LUT gathers.  Only modes that reproduced all 65536 LUT entries
 * bit for bit in the creation-time self-test can be selected; the fastest such mode is the default. */
HYDAMD_EXPORT int hydamd_xyb_mode(HydAmdContext *ctx);
HYDAMD_EXPORT int hydamd_set_xyb_mode(HydAmdContext *ctx, int mode);

/* One of a pixel's six transfer / bias curves is read from the uploaded table through the texture path instead of being
 * evaluated in registers (the values are the reference's own table entries either way): faster for photographic and smooth
 * content, slower for pixels that scatter over the whole table (noise).  0 (default): decided per frame from the density
 * of the frame this context finished last (more than 0.75 bytes of HF sections per pixel: registers); 1: always gather
 * (rounds 1-4); 2: never. */
HYDAMD_EXPORT int hydamd_set_curve_gathers(HydAmdContext *ctx, int mode);
/* The density the default choice goes by is that of the last frame hydamd_sync waited for on this context (bytes and pixels
 * of ONE frame, taken together); hydamd_forget_content clears it — what the drop-in API does when it takes a parked context
 * for another image. */
HYDAMD_EXPORT int hydamd_forget_content(HydAmdContext *ctx);

/* Form of the entropy (rANS) stage.  The recurrence is serial per group, so the forms trade the
 * latency of one frame against how much of the GPU the stage occupies while it runs:
 *   4   one wavefront per group, 4 groups per workgroup (default): lowest single-frame latency
 *   5   one LANE per group: a single wavefront walks the 64 chains of an LF group and a second, wave-
 *       parallel kernel writes the bits straight into the payload; a fifteenth of the instructions and
 *       a sixty-fourth of the wavefronts of form 4 — best when many frames are in flight.  LF groups
 *       with float samples are still coded by form 4.
 *   6   (rounds 3-4: form 5 with its coding tables packed into a byte and a nibble plane, 62 KB of LDS per chain
 *       instead of 80.)  Since round 5 the lane form's tables are sized by the frame's clustering scheme (9 / 3 / 2 / 1
 *       clusters per preset: 79.5 / 26.5 / 17.7 / 8.8 KB) and its step is hand-scheduled; packed and plain tables measured
 *       equal in a pipelined loop, so the packed variant is gone and 6 is accepted as another name of 5.
 * Integer frames: forms 4 and 5 both leave the bits to a second, wave-parallel kernel (k_rans_emit); float frames are
 * coded by form 4's self-emitting variant.
 * Round 1's row forms 1-3 are gone: asking for one of them (here or through HYDAMD_RANS_WAVES) selects
 * form 5, which replaced them, and says so once on stderr. */
HYDAMD_EXPORT int hydamd_set_rans_waves(HydAmdContext *ctx, int waves);

/* Start a frame of `num_presets` presets (= LF groups, at most 255): clears histograms and the running alphabet. */
HYDAMD_EXPORT int hydamd_begin_frame(HydAmdContext *ctx, unsigned num_presets);

/*
 * Enqueue the whole hot path for the LF group stored in `slot` (0 <= slot < max_lf_groups; slots
 * are coded into the frame in the order they are submitted).  src/strides/fmt mean exactly what
 * hyd_send_tile's buffer/row_stride/pixel_stride/sample_fmt mean (strides in samples, src[c]
 * pointing at the LF group's first pixel), except that the pointers are DEVICE pointers.
 */
HYDAMD_EXPORT int hydamd_encode_lf_group(HydAmdContext *ctx, int slot, const void *const src[3], ptrdiff_t row_stride,
                                         ptrdiff_t pixel_stride, int sample_fmt, size_t width, size_t height,
                                         unsigned preset);

/* A whole device-resident image in ONE call: hydamd_begin_frame, hydamd_encode_lf_group for every LF group in raster
 * order (slot = raster index = preset: the one-frame layout of the reference for tiles sent in raster order,
 * libhydrium.c:172-203) and hydamd_finish_frame.  src/strides/fmt as above, for the image's first pixel.  Asynchronous. */
HYDAMD_EXPORT int hydamd_encode_image(HydAmdContext *ctx, const void *const src[3], ptrdiff_t row_stride,
                                      ptrdiff_t pixel_stride, int sample_fmt, size_t width, size_t height);

/* A BATCH of `frames` independent images of one shape as ONE launch group (for a queue of frames: the serial rANS chains
 * of the batch run side by side, so a stream is held for one chain's duration per batch instead of per frame; bench.py
 * codes two 8192x8192 frames per group).  hydamd_begin_batch = hydamd_begin_frame with frame k in slots k * num_presets ...
 * (k + 1) * num_presets - 1, presets 0 .. num_presets - 1 in each; hydamd_encode_image_batch = hydamd_encode_image for
 * src[3 k .. 3 k + 2] of every frame, then hydamd_finish_frame over all slots.  Results per slot as usual
 * (hydamd_read_sections, hydamd_read_tables, hydamd_read_lf_streams); frame k's sections follow frame k - 1's in the
 * payload.  A batch is not exported as a blob (hydamd_export_frame*: HYD_API_ERROR): code one frame per context for that. */
HYDAMD_EXPORT int hydamd_begin_batch(HydAmdContext *ctx, unsigned num_presets, int frames);
HYDAMD_EXPORT int hydamd_encode_image_batch(HydAmdContext *ctx, int frames, const void *const *src, ptrdiff_t row_stride,
                                            ptrdiff_t pixel_stride, int sample_fmt, size_t width, size_t height);

/* Same, from HOST pointers: the samples are gathered into pinned staging (the caller's buffers may
 * be reused as soon as this returns, as after hyd_send_tile) and copied to the GPU on the stream. */
HYDAMD_EXPORT int hydamd_encode_lf_group_host(HydAmdContext *ctx, int slot, const void *const src[3],
                                              ptrdiff_t row_stride, ptrdiff_t pixel_stride, int sample_fmt,
                                              size_t width, size_t height, unsigned preset);

/* Optional: enqueue the transform stage of the LF group in `slot` right away instead of at
 * hydamd_finish_frame (slots in order, one at a time).  hyd_send_tile uses it so that the GPU works
 * on LF group n while the host is still staging tile n + 1; hydamd_finish_frame then runs what is
 * left (the entropy stage and the LF coder, batched over the frame) and packs. */
HYDAMD_EXPORT int hydamd_submit_lf_group(HydAmdContext *ctx, int slot);

/* Optional: enqueue the LF coder for the slots not yet LF-coded below `num_slots` right here in the
 * context's stream (their transform stage must be enqueued); with `last` != 0 the frame's LF streams
 * are also packed, and hydamd_sync_lf() then waits for just that — so a caller can read the LF streams
 * (hydamd_read_lf_streams / hydamd_read_lf_payload) and build the LF sections on the host while
 * the entropy stage enqueued behind it is still running.  hyd_send_tile does. */
HYDAMD_EXPORT int hydamd_run_lf_coder(HydAmdContext *ctx, int num_slots, int last);
HYDAMD_EXPORT int hydamd_sync_lf(HydAmdContext *ctx);

/* Enqueue whatever of the hot path is still outstanding for slots [0, num_slots) — transform stage,
 * LF coder, ANS tables, rANS — and the packing of the HF sections: byte-padded, slot-major, raster
 * inside a slot.  Asynchronous. */
HYDAMD_EXPORT int hydamd_finish_frame(HydAmdContext *ctx, int num_slots);

/* Block until everything enqueued so far has run; reports device-side failures (non-finite float
 * sample -> HYD_API_ERROR, table construction failure -> HYD_INTERNAL_ERROR). */
HYDAMD_EXPORT int hydamd_sync(HydAmdContext *ctx);

/* ---- results; valid after hydamd_sync() ---- */
HYDAMD_EXPORT size_t hydamd_payload_size(HydAmdContext *ctx);
/* The device pointer may change when a frame outgrows its buffers (see below): fetch it after hydamd_sync(). */
HYDAMD_EXPORT const uint8_t *hydamd_payload_device(HydAmdContext *ctx);
/* Buffers are sized for typical content (1.5 symbols and 1 byte of sections per pixel) rather than for
 * the format's hard maximum.  A frame that needs more is detected on the device; hydamd_sync() then
 * enlarges the arrays and runs the frame again before it returns — transparent to the caller except
 * for the time it takes and for device pointers fetched earlier.  These report the current sizes and
 * how often that happened (a context that met one such frame stays enlarged). */
HYDAMD_EXPORT size_t hydamd_payload_capacity(HydAmdContext *ctx);
HYDAMD_EXPORT unsigned hydamd_token_capacity(HydAmdContext *ctx);
HYDAMD_EXPORT unsigned hydamd_overflow_reruns(HydAmdContext *ctx);
/* Once any context of the process has rerun a frame for that reason, the others take the hint: hydamd_begin_frame enlarges
 * an IDLE context's arrays ahead of its next frame (a queue of frames is of one kind as a rule); a busy context keeps what
 * it has and finds out by itself.  How often that happened for this context: */
HYDAMD_EXPORT unsigned hydamd_grown_ahead(HydAmdContext *ctx);
HYDAMD_EXPORT int hydamd_read_payload(HydAmdContext *ctx, uint8_t *dst, size_t capacity);
/* bits[g] = exact bit length of group g's section (0 for absent groups), offsets[g] = byte offset in the payload */
HYDAMD_EXPORT int hydamd_read_sections(HydAmdContext *ctx, int slot, uint32_t bits[HYDAMD_GROUPS_PER_LFG],
                                       uint64_t offsets[HYDAMD_GROUPS_PER_LFG]);
HYDAMD_EXPORT int hydamd_read_tables(HydAmdContext *ctx, int slot, uint32_t freq[HYDAMD_MAX_CLUSTERS][HYDAMD_ALPHABET],
                                     uint32_t alphabet[HYDAMD_MAX_CLUSTERS], uint32_t *log_alphabet_size,
                                     uint32_t *running_max_alphabet);
/* LF ints, dst[c][by][bx] with row pitch vbw, channel order X, Y, B */
HYDAMD_EXPORT int hydamd_read_dc(HydAmdContext *ctx, int slot, int32_t *dst, size_t vbw, size_t vbh);

/* ---- LF-group coder: the LF-coefficient sub-stream of every submitted LF group is coded on the
 * GPU alongside the HF entropy stage (on by default; 0 leaves the LF ints to a host coder via
 * hydamd_read_dc).  Results are valid after hydamd_sync(). ---- */
/* on_device: 0 off; 1 (default) on a side stream beside the HF entropy stage — lowest frame latency;
 * 2 at the end of the context's own stream — no second stream per context, for many frames in flight */
HYDAMD_EXPORT int hydamd_set_lf_coder(HydAmdContext *ctx, int on_device);
HYDAMD_EXPORT int hydamd_lf_coder(HydAmdContext *ctx);
/* code length per compact token, largest token + 1, number of (run, distance) pairs, bits of symbol data */
HYDAMD_EXPORT int hydamd_read_lf_stream(HydAmdContext *ctx, int slot, uint8_t lengths[HYDAMD_LF_CODES], uint32_t *alphabet,
                                        uint32_t *run_pairs, uint32_t *bit_count);
/* the symbol bits, LSB first; capacity >= (bit_count + 7) / 8 */
HYDAMD_EXPORT int hydamd_read_lf_bits(HydAmdContext *ctx, int slot, uint8_t *dst, size_t capacity);
/* The frame's LF streams in two copies instead of two per slot: the per-slot records (with the byte
 * offset of each slot's symbol data) and the symbol data of all slots back to back, 4-byte aligned,
 * in slot order.  hydamd_lf_payload_device() is what a multi-GPU job all-gathers. */
typedef struct HydAmdLfInfo {
    uint32_t bit_count, alphabet, run_pairs, error, offset, reserved[3];
    uint8_t lengths[HYDAMD_LF_CODES];
} HydAmdLfInfo;
HYDAMD_EXPORT int hydamd_read_lf_streams(HydAmdContext *ctx, int first_slot, int count, HydAmdLfInfo *dst);
HYDAMD_EXPORT size_t hydamd_lf_payload_size(HydAmdContext *ctx);
HYDAMD_EXPORT const uint8_t *hydamd_lf_payload_device(HydAmdContext *ctx);
HYDAMD_EXPORT int hydamd_read_lf_payload(HydAmdContext *ctx, uint8_t *dst, size_t capacity);
/* unit-test entry: run the device's code construction on one histogram over the compact token space */
HYDAMD_EXPORT int hydamd_debug_lf_code(HydAmdContext *ctx, const uint32_t hist[HYDAMD_LF_CODES],
                                       uint8_t lengths[HYDAMD_LF_CODES], uint32_t codes[HYDAMD_LF_CODES],
                                       uint32_t *alphabet, uint32_t *error);

/* Resource footprint of the transform kernel instance that serves `sample_fmt` on this context (the XYB mode the
 * context runs): static LDS bytes and registers per thread.  LDS is allocated in granules of 1280 bytes and two
 * transform workgroups share a CU with one entropy-stage workgroup only at <= 26 granules: a test holds that line. */
HYDAMD_EXPORT int hydamd_debug_transform_footprint(HydAmdContext *ctx, int sample_fmt, int *lds_bytes, int *registers);
/* the shader clock in MHz as a 20 us single-wavefront kernel on a stream of its own sees it (s_memtime against the 100 MHz
 * s_memrealtime); blocks the caller for those 20 us plus the launch, not the context's stream */
HYDAMD_EXPORT int hydamd_debug_shader_clock_mhz(HydAmdContext *ctx, double *mhz);

/* ---- parity / debug read-backs ---- */
HYDAMD_EXPORT int hydamd_read_symbol_counts(HydAmdContext *ctx, int slot, uint32_t counts[HYDAMD_GROUPS_PER_LFG]);
/* token records of one group: lo = token | cluster<<8 | residue_bits<<16, hi = residue */
HYDAMD_EXPORT int hydamd_read_tokens(HydAmdContext *ctx, int slot, int group, uint64_t *dst, size_t capacity);
/* which: 0 XYB, 1 DCT (float), 2 quantised (int32); dst[c][y][x] with row pitch `pitch` elements */
HYDAMD_EXPORT int hydamd_read_debug_plane(HydAmdContext *ctx, int which, void *dst, size_t pitch, size_t rows);

/* ---- multi-GPU: only the entropy tables of LF group n depend on earlier LF groups, through the
 * running maximum alphabet (reference entropy.c:459-460,952).  A rank that codes LF groups
 * k..k+m of a frame runs the transform stage, exchanges the per-LF-group maxima, sets the maximum
 * over LF groups 0..k-1 as its floor, then runs the entropy stage.  hydamd_finish_frame() is the
 * two stages back to back with floor 0. ---- */
HYDAMD_EXPORT int hydamd_run_transform(HydAmdContext *ctx, int num_slots);
HYDAMD_EXPORT int hydamd_read_alphabet_max(HydAmdContext *ctx, int slot, uint32_t *max_token_plus_one);
HYDAMD_EXPORT int hydamd_set_alphabet_floor(HydAmdContext *ctx, uint32_t floor);
HYDAMD_EXPORT int hydamd_run_entropy(HydAmdContext *ctx, int num_slots);

/* The same exchange without the host in the loop: the per-slot maxima as they sit in device memory
 * ([max_lf_groups] uint32, complete once the transform kernels enqueued so far have run — order your
 * reads behind the context's stream), and a device location the table kernel reads the floor from when
 * it runs (the larger of it and hydamd_set_alphabet_floor's value counts; NULL clears; reset by
 * hydamd_begin_frame).  A job that all-gathers the maxima with RCCL writes its floor there. */
HYDAMD_EXPORT const uint32_t *hydamd_alphabet_max_device(HydAmdContext *ctx);
HYDAMD_EXPORT int hydamd_set_alphabet_floor_device(HydAmdContext *ctx, const uint32_t *floor_on_device);

/* One frame on several devices of ONE process (what hyd_send_tile does when HYDAMD_DEVICES names more than one; no
 * collective library involved): every device's context owns a run of consecutive LF groups in send order.
 *   hydamd_wait_for                   ctx's stream waits, on the device, for everything enqueued so far on peer's
 *                                     stream; peer access from ctx's device to peer's is enabled (xGMI reads).
 *   hydamd_alphabet_floor_from_peers  the floor of ctx's LF groups = the largest token + 1 over the LF groups of the
 *                                     `npeers` contexts that hold the LF groups sent BEFORE ctx's (their transform
 *                                     stages enqueued): a single-wave kernel in ctx's stream reads the peers' maxima
 *                                     in place, behind their transform kernels, and leaves the floor where ctx's
 *                                     table kernel reads it.  Then hydamd_run_entropy / hydamd_finish_frame.
 * The assembler (below) reads blobs of other devices in place: hydamd_wait_for(assembling ctx, peer) first. */
#define HYDAMD_MAX_PEERS 8
HYDAMD_EXPORT int hydamd_context_device(HydAmdContext *ctx);
HYDAMD_EXPORT int hydamd_wait_for(HydAmdContext *ctx, HydAmdContext *peer);
/* 1 if every device of the list can read every other one's memory (hipDeviceCanAccessPeer for every ordered pair of distinct
 * devices; an index that repeats is its own peer), else 0.  hyd_send_tile asks before it deals a frame out to several
 * devices and keeps the frame on one device when the answer is no. */
HYDAMD_EXPORT int hydamd_peers_reachable(const int *devices, int n);
/* Insurance for the peer reads (what HYDAMD_VERIFY_PEERS=1 makes hyd_send_tile do for every sharded frame): a checksum of
 * everything `owner`'s exported view names (view, packed LF streams, HF sections), computed in `reader`'s stream — on the
 * owning device when reader == owner, through peer reads otherwise (hydamd_wait_for(reader, owner) first) — into the
 * reader's result `index` (0 .. HYDAMD_MAX_PEERS - 1); hydamd_verify_read waits for the reader's stream and returns it.
 * The owner's and a reader's sums of one view are equal exactly when the reader saw the bytes the owner wrote. */
HYDAMD_EXPORT int hydamd_verify_enqueue(HydAmdContext *reader, HydAmdContext *owner, int num_slots, int index);
HYDAMD_EXPORT int hydamd_verify_read(HydAmdContext *reader, int index, unsigned long long *sum);
HYDAMD_EXPORT int hydamd_alphabet_floor_from_peers(HydAmdContext *ctx, int npeers, HydAmdContext *const *peers);
/* The other peer read of a sharded frame, checked the same way: *ok = 1 when the floor hydamd_alphabet_floor_from_peers left
 * for ctx's table kernel equals the maximum over the peers' per-LF-group maxima as the HOST copies them from each peer's own
 * device (no peer access involved).  Waits for the frames of ctx and of the peers. */
HYDAMD_EXPORT int hydamd_verify_floor(HydAmdContext *ctx, int npeers, HydAmdContext *const *peers, int *ok);
/* Enqueue the current frame's stages again from the job descriptors the context still holds (pixels stay borrowed), after
 * waiting for what was enqueued before — for a caller that changed an input of the closing stage (hydamd_set_alphabet_floor,
 * hydamd_set_alphabet_floor_device) after hydamd_finish_frame.  hyd_send_tile's through-the-host fallback uses it when a
 * sharded frame's peer reads fail their first-use verification.  Not for batches. */
HYDAMD_EXPORT int hydamd_replay_frame(HydAmdContext *ctx);

/*
 * One self-describing byte string with everything a frame assembler needs from this context's LF
 * groups, built by a kernel in the context's stream right behind the entropy stage: no host
 * synchronisation, one buffer to gather.  Layout: HydAmdBlobHeader, num_slots x HydAmdBlobSlot, the
 * packed LF streams (lf_bytes), then at the next multiple of 16 the packed HF sections (hf_bytes).
 * The LF coder must be on.  `capacity` is the size of the caller's device buffer; a blob that does not
 * fit, or a frame that outgrew the context's buffers, has HYDAMD_BLOB_RETRY set in header.status: call
 * hydamd_sync() (it enlarges and reruns the frame) and export again, into a larger buffer if
 * total_bytes says so.  hydamd_blob_bound() is an upper bound for the context's current capacities.
 */
typedef struct HydAmdBlobHeader {
    uint32_t magic;       /* "HYDB" */
    uint32_t version;     /* 1 */
    uint32_t num_slots;
    uint32_t status;      /* device status word; & HYDAMD_BLOB_RETRY: incomplete, see above; & 1: non-finite float sample */
    uint64_t hf_bytes, lf_bytes, total_bytes;
    uint32_t lf_coded;    /* the slot records carry device-coded LF streams */
    uint32_t reserved[5];
} HydAmdBlobHeader;
#define HYDAMD_BLOB_RETRY 0xEu
typedef struct HydAmdBlobSlot {
    uint32_t preset;                /* = raster id of the LF group in its frame */
    uint32_t running_max_alphabet, log_alphabet_size, table_error;
    uint32_t alphabet[HYDAMD_MAX_CLUSTERS];
    uint32_t reserved[3];
    uint32_t group_bits[HYDAMD_GROUPS_PER_LFG];
    uint32_t freq[HYDAMD_MAX_CLUSTERS][HYDAMD_ALPHABET];
    HydAmdLfInfo lf;                /* lf.offset is relative to the blob's LF byte string */
} HydAmdBlobSlot;
HYDAMD_EXPORT size_t hydamd_blob_bound(HydAmdContext *ctx, int num_slots);
HYDAMD_EXPORT int hydamd_export_frame(HydAmdContext *ctx, int num_slots, void *device_dst, size_t capacity);
/* The same blob through a device buffer and a pinned host buffer of the context's own: stage (enqueued on the context's
 * stream, behind hydamd_finish_frame), hydamd_sync, read — ONE device-to-host copy of exactly the blob's bytes; *host_blob
 * stays valid until the context's next staged read; *size = 0 (and a blob whose header carries HYDAMD_BLOB_RETRY) when the frame
 * was rerun with larger buffers after it was staged: read the results the separate way then.  What hyd_send_tile uses for the frames it assembles on the host
 * (tile mode): six small read-backs became one. */
HYDAMD_EXPORT int hydamd_stage_frame_blob(HydAmdContext *ctx, int num_slots);
HYDAMD_EXPORT int hydamd_read_frame_blob(HydAmdContext *ctx, int num_slots, const void **host_blob, size_t *size);
/* Host only: a whole one-frame codestream from the blobs of the contexts (ranks) that coded its LF
 * groups, in any order; every LF group of the image must appear exactly once.  *out as for
 * hydamd_frame_from_streams. */
HYDAMD_EXPORT int hydamd_frame_from_blobs(const HYDImageMetadata *md, int write_header, int is_last, size_t nblobs,
                                          const void *const *blobs, const size_t *blob_sizes, const uint8_t *icc,
                                          size_t icc_size, uint8_t **out, size_t *out_len, const char **err);

/*
 * The same assembly ON THE DEVICE: the blobs stay where hydamd_export_frame (or an RCCL gather) left
 * them, kernels write every section into place — LF group sections around their coefficient streams
 * (reference encoder.c:539-629), HFGlobal's histograms (encoder.c:959-967), the TOC (encoder.c:992-1005)
 * — and the finished one-frame codestream lands in one buffer: one copy to the host instead of
 * per-section read-backs and a host-side splice.  The host contributes the bytes that do not depend on
 * the pixels (file and frame header, LFGlobal, constant sub-streams), once per frame description.
 *   hydamd_assembler_plan    describe the frame: which LF groups (raster ids, `lf_ids`) each of the
 *                            `nblobs` blobs carries, blob by blob, slot by slot — this order becomes the
 *                            frame's section order.  Cheap when the description repeats.
 *   hydamd_assembler_run     enqueue the assembly on `hip_stream` (behind whatever fills the blobs):
 *                            `blobs_dev[b]` is a device pointer to blob b (`blob_caps[b]` readable bytes),
 *                            `out` any device-accessible buffer of `out_cap` bytes (device memory, or
 *                            pinned host memory for frames that should land on the host directly).
 *                            Alignment: `out` 4 bytes, every blob 16 bytes (the copy kernel moves whole
 *                            words and 16-byte records); anything else is HYD_API_ERROR.
 *   hydamd_assembler_result  after the stream has been synchronised: the frame's size; HYD_NEED_MORE_OUTPUT
 *                            (with *size = bytes needed) if `out_cap` was too small; HYD_API_ERROR for a
 *                            blob that is malformed, incomplete (rerun the shard) or carries NaN input.
 * Frames of a single 256x256 group are one bit-contiguous section and stay with hydamd_frame_from_blobs.
 * One assembler serves one frame at a time (its scratch is reused by the next hydamd_assembler_run).
 */
typedef struct HydAmdAssembler HydAmdAssembler;
HYDAMD_EXPORT HydAmdAssembler *hydamd_assembler_create(int device, int *status);
HYDAMD_EXPORT void hydamd_assembler_destroy(HydAmdAssembler *a);
HYDAMD_EXPORT const char *hydamd_assembler_error(HydAmdAssembler *a);
HYDAMD_EXPORT int hydamd_assembler_plan(HydAmdAssembler *a, const HYDImageMetadata *md, int write_header, int is_last,
                                        size_t nblobs, const uint32_t *blob_slots, const uint32_t *lf_ids, const uint8_t *icc,
                                        size_t icc_size);
HYDAMD_EXPORT int hydamd_assembler_run(HydAmdAssembler *a, const void *const *blobs_dev, const size_t *blob_caps, void *hip_stream,
                                       void *out, size_t out_cap);
HYDAMD_EXPORT int hydamd_assembler_result(HydAmdAssembler *a, size_t *size);
/* `out` == NULL in hydamd_assembler_run: the frame goes to a device buffer the assembler owns — of `out_cap` bytes, or,
 * with out_cap == 0, sized from the blob capacities (self-contained blobs only: a view's capacity says nothing about
 * its frame; hydamd_blob_bound() of the exporting context is an upper bound for it) — and this copies it to host
 * memory once hydamd_assembler_result has said how large it is. */
HYDAMD_EXPORT int hydamd_assembler_read(HydAmdAssembler *a, uint8_t *dst, size_t capacity);
/* A context's own way to both: the blob of slots [0, num_slots) as a VIEW — header and slot records in a small device
 * buffer the context keeps, the packed LF streams and HF sections left where the context has them, their addresses in
 * the header (lf_coded = 0x101) — for an assembler on the same device and stream, valid until the context's next frame:
 * nothing of the frame's bulk is copied.  hydamd_frame_from_blobs does not take views.  *capacity receives the readable
 * bytes at *blob_dev.  And an assembler that lives and is parked with the context.  hyd_send_tile builds its frames
 * with these. */
HYDAMD_EXPORT int hydamd_export_frame_owned(HydAmdContext *ctx, int num_slots, const void **blob_dev, size_t *capacity);
HYDAMD_EXPORT HydAmdAssembler *hydamd_context_assembler(HydAmdContext *ctx);

/*
 * Wrap LF-group results — from this or other GPUs — into codestream bytes (host only, no GPU).
 *   md            image metadata, as for hyd_set_metadata (one-frame mode: every LF group must be present)
 *   tile_xy       [lfg_count][2] tile coordinates, in the order the sections appear in `payload`
 *   dc            [lfg_count] pointers to LF ints as hydamd_read_dc returns them
 *   freq/alphabet [lfg_count][9][128] / [lfg_count][9] as hydamd_read_tables returns them
 *   group_bits    [lfg_count][64] as hydamd_read_sections returns them
 *   max_alphabet  final running maximum (largest running_max_alphabet of any LF group)
 *   payload       the packed HF sections of all LF groups, concatenated in the same order
 * On success *out is a malloc'ed buffer (release with hydamd_free) holding the file header (if
 * write_header) and the frame.
 */
HYDAMD_EXPORT int hydamd_frame_from_results(const HYDImageMetadata *md, int write_header, int is_last, size_t lfg_count,
                                            const uint32_t *tile_xy, const int32_t *const *dc, const uint32_t *freq,
                                            const uint32_t *alphabet, const uint32_t *group_bits, unsigned max_alphabet,
                                            const uint8_t *payload, size_t payload_len, const uint8_t *icc,
                                            size_t icc_size, uint8_t **out, size_t *out_len, const char **err);
/* The same with LF groups whose coefficient streams were coded on the GPU (hydamd_read_lf_stream /
 * hydamd_read_lf_bits) instead of LF ints: what a multi-GPU job gathers when the LF coder is on. */
typedef struct HydAmdLfStream {
    const uint8_t *lengths;  /* [HYDAMD_LF_CODES] */
    uint32_t alphabet, run_pairs;
    const uint8_t *bits;     /* (bit_count + 7) / 8 bytes */
    uint64_t bit_count;
} HydAmdLfStream;
HYDAMD_EXPORT int hydamd_frame_from_streams(const HYDImageMetadata *md, int write_header, int is_last, size_t lfg_count,
                                            const uint32_t *tile_xy, const HydAmdLfStream *lf, const uint32_t *freq,
                                            const uint32_t *alphabet, const uint32_t *group_bits, unsigned max_alphabet,
                                            const uint8_t *payload, size_t payload_len, const uint8_t *icc,
                                            size_t icc_size, uint8_t **out, size_t *out_len, const char **err);
/* Releases a buffer returned by hydamd_frame_from_*.  The library may keep it for the next frame it
 * assembles (mapped pages: a fresh 50 MB buffer costs 8 ms of page faults): up to four buffers of 1 MB to
 * 256 MB each per process, i.e. at most 1 GB of host memory; the next frame or encoder takes the smallest
 * one that fits it.  The buffer behind *out may therefore be larger than out_len.  hydamd_trim_cache()
 * lets go of all of them. */
HYDAMD_EXPORT void hydamd_free(void *p);

/* hyd_encoder_destroy parks its device context (device memory, pinned staging, streams) for the next
 * encoder of the same shape instead of freeing it — up to HYDAMD_CONTEXT_CACHE contexts (default 16, at most 32) and
 * HYDAMD_CONTEXT_CACHE_MB megabytes (default 8192) per process.  This releases whatever is parked, and the spare
 * frame buffers hydamd_free may have kept. */
HYDAMD_EXPORT void hydamd_trim_cache(void);

/* ---- knobs of the drop-in encoder (HYDEncoder of libhydrium.h) that the reference has no counterpart for ----
 * Tile mode (tile_size_shift >= 0): every tile is a frame, and by default — as in the reference
 * (src/libhydrium/libhydrium.c:147-203, encoder.c:339-378) — hyd_send_tile returns with that frame complete.
 * hydamd_set_tile_pipeline(e, depth) lets up to `depth` (2..8) tile frames be in flight instead: a call launches
 * its tile and collects the frame launched `depth` calls earlier, the final tile's call collects the rest; the
 * bytes are the same, in the same order, up to depth - 1 calls later (4-5x the tile rate: a tile costs 2-3 ms of
 * latency however small it is), a NaN or device error is reported by the call that collects the frame, and an
 * encoder destroyed before its final tile drops what is still in flight.  depth 1 = the reference's timing;
 * 0 = the process default (environment HYDAMD_TILE_PIPELINE, else 1).  HYD_API_ERROR while frames are in flight. */
HYDAMD_EXPORT int hydamd_set_tile_pipeline(HYDEncoder *encoder, int depth);
HYDAMD_EXPORT int hydamd_get_tile_pipeline(const HYDEncoder *encoder);

/* ---- optional per-kernel timing with HIP events on the context's stream ---- */
HYDAMD_EXPORT int hydamd_profile(HydAmdContext *ctx, int enable);
/* accumulated milliseconds and launch counts per kernel class since the last call; resets the counters */
HYDAMD_EXPORT int hydamd_profile_read(HydAmdContext *ctx, double ms[HYDAMD_K_COUNT], uint64_t launches[HYDAMD_K_COUNT]);

/*
 * ONE frame whose pixels already sit in HBM, on N devices of ONE process (csrc/host/multi.c; the composition of the calls
 * above that hyd_send_tile's multi-device scheduler makes for host tiles — reference libhydrium.c:172-203,
 * encoder.c:928-957 — without the uploads, which bound that path at any N).  No collective library, no process group:
 * LF groups dealt in raster runs (shard d: groups d * total / N .. (d + 1) * total / N - 1), the alphabet floor by peer
 * read, every shard's blob a view, the assembling shard's GPU reading all of them in place (the other devices' over xGMI)
 * and writing the finished FILE (header included) into its own memory.
 *   hydamd_multi_create        contexts for the image's shape, one per entry of `devices` (an index may repeat: several
 *                              shards on one GPU); HYD_INTERNAL_ERROR in *status when the devices cannot read each other.
 *   hydamd_encode_image_multi  enqueue one frame; returns without waiting.  `src` holds three pointers per shard
 *                              ([shard][channel]) in THAT shard's device memory: where pixel (0, 0) of the image would sit
 *                              in the shard's buffer — only the pixels of the shard's own LF groups are read, so a slab of
 *                              rows [y0, y1) passes slab - y0 * row_stride.  Strides in samples, sample_fmt as
 *                              hyd_send_tile's.  `assembling_shard` picks the GPU that builds the file (rotate it per frame:
 *                              the file's copy to the host then leaves through another GPU's link each time).
 *   hydamd_multi_result        wait for the frame; reruns what a shard that outgrew its buffers invalidated (its own frame
 *                              inside hydamd_sync, the later shards' floors and frames, the assembly).  *size = bytes of
 *                              the file.  A peer read that fails its first-use verification (see HYDAMD_VERIFY_PEERS in
 *                              INTEGRATION.md) is HYD_INTERNAL_ERROR with the device pair in hydamd_multi_error.
 *   hydamd_multi_read          the file to host memory, one copy from the assembling device.
 * One frame in flight per HydAmdMulti; keep several for a queue of frames (bench.py's shard_16k_inprocess leg keeps four).
 */
typedef struct HydAmdMulti HydAmdMulti;
HYDAMD_EXPORT HydAmdMulti *hydamd_multi_create(int n, const int *devices, const HYDImageMetadata *md, int *status);
HYDAMD_EXPORT void hydamd_multi_destroy(HydAmdMulti *m);
HYDAMD_EXPORT const char *hydamd_multi_error(HydAmdMulti *m);
HYDAMD_EXPORT HydAmdContext *hydamd_multi_context(HydAmdMulti *m, int shard);
HYDAMD_EXPORT int hydamd_multi_shard_lf_groups(HydAmdMulti *m, int shard, size_t *first, size_t *count);
HYDAMD_EXPORT int hydamd_encode_image_multi(HydAmdMulti *m, const void *const *src, ptrdiff_t row_stride, ptrdiff_t pixel_stride,
                                            int sample_fmt, int assembling_shard);
HYDAMD_EXPORT int hydamd_multi_result(HydAmdMulti *m, size_t *size);
HYDAMD_EXPORT int hydamd_multi_read(HydAmdMulti *m, uint8_t *dst, size_t capacity);

#ifdef __cplusplus
}
#endif

#endif /* HYDRIUM_AMD_H_ */
