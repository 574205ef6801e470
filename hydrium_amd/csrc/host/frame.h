/*
 * frame.h — host-side frame glue: everything that wraps the GPU-coded HF group sections into a
 * valid JPEG XL codestream (reference encoder.c "frame glue").  These are
 * O(1) .. O(varblocks) bit-field writers; they stay on the host by design.
 */
#ifndef HYD_FRAME_H_
#define HYD_FRAME_H_

#include <stddef.h>
#include <stdint.h>

#include "bitio.h"

#define HYD_FRAME_MAX_CLUSTERS 9
#define HYD_FRAME_ALPHABET 128

/* One LF group of the frame, in the order the caller sent it. */
typedef struct HydFrameLfg {
    size_t raster_id;     /* y * lf_groups_across + x (0 in tile mode) */
    size_t x, y;          /* LF-group coordinates, in LF groups (tile coordinates in tile mode) */
    size_t width, height; /* pixels */
} HydFrameLfg;

typedef struct HydFrameShape {
    int one_frame;
    size_t image_width, image_height;
    size_t frame_width, frame_height; /* the whole image in one-frame mode, the tile otherwise */
    size_t tile_count_x, tile_count_y; /* groups per tile side (tile mode crop origin) */
    size_t lfg_count;                 /* LF groups in this frame */
    const HydFrameLfg *lfg;           /* [lfg_count] in send order */
    int is_last;                      /* last frame of the file */
} HydFrameShape;

/* file signature + SizeHeader + ImageMetadata (+ ICC stream), byte-padded (encoder.c:164-239) */
int hyd_write_file_header(HydBits *out, size_t width, size_t height, int level10, const uint8_t *icc, size_t icc_size,
                          const char **err);
/* FrameHeader + permuted-TOC flag and Lehmer code, byte-padded (encoder.c:241-435) */
int hyd_write_frame_header(HydBits *out, const HydFrameShape *shape, const char **err);
/* number of TOC entries of the frame (encoder.c:281) */
size_t hyd_toc_entries(const HydFrameShape *shape);
/* TOC section sizes, byte-padded (encoder.c:992-1005) */
int hyd_write_toc_sizes(HydBits *out, const size_t *section_bytes, size_t count);

void hyd_write_lf_global(HydBits *out);                                           /* encoder.c:510-537 */
/* dc[c][by][bx] with row pitch vbw, channels X, Y, B (encoder.c:539-629) */
int hyd_write_lf_group(HydBits *out, const int32_t *dc, size_t vbw, size_t vbh, const char **err);

/* The same section when the LF-coefficient stream was coded on the GPU (csrc/hip/lf_coder.hip): the
 * host writes the constant sub-streams and the code-length header around the device's symbol bits.
 * lengths[] is indexed by compact token: [0,256) literal tokens, [256,384) token 16384 + (i - 256). */
#define HYD_LF_RUN_BASE 16384u
#define HYD_LF_CODES 384
typedef struct HydLfCoded {
    const uint8_t *lengths;   /* [HYD_LF_CODES] */
    uint32_t alphabet;        /* largest token + 1 of the value cluster */
    uint32_t run_pairs;       /* (run token, distance) pairs in the stream */
    const uint8_t *bits;      /* symbol bits, LSB first */
    uint64_t bit_count;
} HydLfCoded;
/* `tail`: optional result of hyd_write_lf_group_tail for this (vbw, vbh) — the HF-metadata
 * sub-streams depend on the LF group's geometry only, so a frame needs them coded at most four times */
int hyd_write_lf_group_coded(HydBits *out, size_t vbw, size_t vbh, const HydLfCoded *lf, const HydBits *tail,
                             const char **err);
int hyd_write_lf_group_tail(HydBits *out, size_t vbw, size_t vbh, const char **err);
/* the bits of an LF group section in front of its first data-dependent one (device-side assembler) */
int hyd_write_lf_group_fixed_head(HydBits *out, const char **err);

/* HFGlobal's geometry-only fields, up to and including "ANS, not prefix codes" (device-side assembler) */
int hyd_write_hf_global_fixed(HydBits *out, unsigned num_presets, size_t num_frame_groups, int *clusters_per_preset,
                              const char **err);
/* HF context -> cluster map of a frame with num_presets presets (encoder.c:852-901); returns clusters per preset */
int hyd_hf_cluster_map(uint8_t *map, unsigned num_presets);

/*
 * HFGlobal (encoder.c:959-967 + entropy.c:980-1001, 303-369, 546-575).
 *   freq / alphabet: per preset, the normalised tables of its clusters as the device produced them
 *   max_alphabet:    final running maximum of token + 1 over the whole frame
 */
int hyd_write_hf_global(HydBits *out, unsigned num_presets, size_t num_frame_groups,
                        const uint32_t (*freq)[HYD_FRAME_MAX_CLUSTERS][HYD_FRAME_ALPHABET],
                        const uint32_t (*alphabet)[HYD_FRAME_MAX_CLUSTERS], unsigned max_alphabet, const char **err);

#endif /* HYD_FRAME_H_ */
