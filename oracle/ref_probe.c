/*
 * oracle/ref_probe.c — TEST INFRASTRUCTURE ONLY (container-only; needs /root/reference).
 *
 * Stage-level window into the real reference encoder, used to pin oracle/hyd_oracle.c and to
 * generate the fixtures under tests/golden/.  It pulls the reference's encoder.c in by path
 * (-DREF_ENCODER_C=...) so the file-static forward_dct (encoder.c:631) is reachable, and it
 * reads HYDEncoder's private fields through the reference's own internal.h.  No reference source
 * is copied into this repository; this file only names the reference's symbols.
 *
 * Driving pattern (see tests/refprobe.py): the caller uses the normal public API with
 * is_last = 0, which leaves every intermediate of the LF group alive after hyd_send_tile
 * (encoder.c:956 returns before the frame is finalised):
 *   - encoder->xyb          quantised HF ints + LF ints (encoder.c:582,808-812)
 *   - hf_stream.symbols     the LF group's tokens (symbol_count is reset, memory is not)
 *   - hf_stream_barrier[g]  per-group symbol counts
 *   - hf_stream.frequencies normalised ANS tables
 *   - hf_coeffs[g]          per-group bitstreams
 */
#include REF_ENCODER_C

#include "format.h"

#define PROBE __attribute__((visibility("default")))

/* ---- staged execution ------------------------------------------------------------------ */

/* hyd_send_tile (libhydrium.c:172-203) up to and including the XYB conversion only. */
PROBE int refp_stage_xyb(HYDEncoder *enc, const void *const buffer[3], uint32_t tile_x, uint32_t tile_y,
                         ptrdiff_t row_stride, ptrdiff_t pixel_stride, int is_last, int fmt) {
    HYDStatusCode ret = hyd_send_tile_pre(enc, tile_x, tile_y, is_last);
    if (ret < HYD_ERROR_START)
        return ret;
    size_t lfid = enc->one_frame ? tile_y * enc->lfg_count_x + tile_x : 0;
    return hyd_populate_xyb_buffer(enc, buffer, row_stride, pixel_stride, lfid, (HYDSampleFormat)fmt);
}

/* forward_dct (encoder.c:631-668) on the LF group currently held in encoder->xyb. */
PROBE void refp_stage_dct(HYDEncoder *enc, size_t lfid) {
    forward_dct(enc, &enc->lfg[lfid]);
}

/* ---- accessors ------------------------------------------------------------------------- */

PROBE size_t refp_sizeof_encoder(void) { return sizeof(HYDEncoder); }

PROBE const void *refp_xyb(HYDEncoder *enc) { return enc->xyb; }

/* out[0..8] = tile_count_x, tile_count_y, x, y, width, height, varblock_width, varblock_height, stride */
PROBE void refp_lfg(HYDEncoder *enc, size_t lfid, size_t out[9]) {
    const HYDLFGroup *g = &enc->lfg[lfid];
    out[0] = g->tile_count_x; out[1] = g->tile_count_y; out[2] = g->x; out[3] = g->y;
    out[4] = g->width; out[5] = g->height; out[6] = g->varblock_width; out[7] = g->varblock_height;
    out[8] = g->stride;
}

PROBE size_t refp_groups_encoded(HYDEncoder *enc) { return enc->groups_encoded; }
PROBE size_t refp_num_frame_groups(HYDEncoder *enc) { return enc->num_hf_coeff_bw; }

PROBE size_t refp_barrier(HYDEncoder *enc, size_t g, int *preset) {
    if (!enc->hf_stream_barrier)
        return 0;
    if (preset)
        *preset = enc->hf_stream_barrier[g].preset;
    return enc->hf_stream_barrier[g].barrier_index;
}

/* tokens: 8-byte records {u16 token, u8 cluster, u8 residue_bits, u32 residue} (entropy.h:9-14) */
PROBE const void *refp_symbols(HYDEncoder *enc) { return enc->hf_stream.symbols; }
PROBE size_t refp_symbol_capacity(HYDEncoder *enc) { return enc->hf_stream.symbol_capacity; }
PROBE size_t refp_num_clusters(HYDEncoder *enc) { return enc->hf_stream.num_clusters; }
PROBE int refp_max_alphabet_size(HYDEncoder *enc) { return enc->hf_stream.max_alphabet_size; }
PROBE int refp_alphabet_size(HYDEncoder *enc, size_t cluster) { return enc->hf_stream.alphabet_sizes[cluster]; }
PROBE const uint32_t *refp_frequencies(HYDEncoder *enc, size_t cluster) { return enc->hf_stream.frequencies[cluster]; }
PROBE const uint8_t *refp_cluster_map(HYDEncoder *enc, size_t *n) {
    if (n)
        *n = enc->hf_stream.num_dists;
    return enc->hf_stream.cluster_map;
}

/* per-group HF bitstream: whole bytes in buffer[0..pos) plus cache_bits (<64) pending bits in cache */
PROBE const uint8_t *refp_hf_coeff(HYDEncoder *enc, size_t g, size_t *pos, uint64_t *cache, int *cache_bits) {
    if (!enc->hf_coeffs)
        return NULL;
    const HYDBitWriter *bw = &enc->hf_coeffs[g];
    *pos = bw->buffer_pos;
    *cache = bw->cache;
    *cache_bits = bw->cache_bits;
    return bw->buffer;
}

/* the three LUTs the integer path builds (format.c:58-83) */
PROBE const uint16_t *refp_input_lut8(HYDEncoder *enc) { return enc->input_lut8; }
PROBE const uint16_t *refp_input_lut16(HYDEncoder *enc) { return enc->input_lut16; }
PROBE const float *refp_bias_lut(HYDEncoder *enc) { return enc->bias_cbrtf_lut; }

/* the frame-glue sections accumulated so far in working_writer (LFGlobal, LFGroups, ...) */
PROBE const uint8_t *refp_working(HYDEncoder *enc, size_t *pos, uint64_t *cache, int *cache_bits) {
    *pos = enc->working_writer.buffer_pos;
    *cache = enc->working_writer.cache;
    *cache_bits = enc->working_writer.cache_bits;
    return enc->working_writer.buffer;
}
PROBE size_t refp_section_endpos(HYDEncoder *enc, size_t idx) {
    return enc->section_endpos ? enc->section_endpos[idx] : 0;
}
PROBE size_t refp_section_count(HYDEncoder *enc) { return enc->section_count; }
