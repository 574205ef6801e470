/*
 * prefix.h — the prefix-code (Brotli-style Huffman) entropy coder of the frame glue.
 *
 * Used for every stream that is NOT on the GPU hot path: the TOC permutation, the LF-group
 * modular streams, the nested cluster map of HFGlobal and the ICC profile.  It restates, with the
 * same bit-level results, the reference's hybrid-uint + LZ77-as-RLE symbol buffering
 * (entropy.c:371-524), its depth-limited Huffman construction with the reference's exact
 * tie-breaking (entropy.c:577-662), canonical code assignment (entropy.c:664-707) and code-length
 * coding (entropy.c:709-941), plus the stream-header fields shared with the ANS coder
 * (entropy.c:108-182,546-575).
 */
#ifndef HYD_PREFIX_H_
#define HYD_PREFIX_H_

#include <stddef.h>
#include <stdint.h>

#include "bitio.h"

typedef struct HydUintConfig {
    uint8_t split_exponent, msb_in_token, lsb_in_token;
} HydUintConfig;

typedef struct HydSym {
    uint16_t token;
    uint8_t cluster;
    uint8_t residue_bits;
    uint32_t residue;
} HydSym;

typedef struct HydSymStream {
    size_t num_dists;         /* contexts, including the LZ77 distance context when RLE is on */
    uint8_t *cluster_map;     /* [num_dists] */
    size_t num_clusters;
    HydUintConfig config[256];
    uint16_t alphabet[256];   /* largest token + 1 per cluster */
    uint16_t max_alphabet;
    HydSym *sym;
    size_t count, cap;
    /* LZ77 used as run-length coding of repeated symbols (entropy.c:473-524) */
    uint32_t rle_min_symbol;  /* 0: off */
    uint32_t rle_min_length;
    uint32_t last_value_plus1;
    uint32_t last_dist;
    uint32_t run;
    int modular;
    int failed;
} HydSymStream;

/* status: 0 ok, HYD_NOMEM / HYD_INTERNAL_ERROR values otherwise; *err receives a static message */
int hps_init(HydSymStream *s, const uint8_t *cluster_map, size_t num_dists, int custom_configs,
             uint32_t rle_min_symbol, int modular);
void hps_set_config(HydSymStream *s, uint8_t from_cluster, uint8_t to_cluster, int split_exponent, int msb, int lsb);
int hps_send(HydSymStream *s, size_t dist, uint32_t value);
/* header + all symbols, then frees the stream (hyd_prefix_finalize_stream, entropy.c:1023-1034) */
int hps_finish_prefix(HydSymStream *s, HydBits *out, const char **err);
void hps_free(HydSymStream *s);

/* The part of a prefix-coded stream in front of its symbols, for callers whose symbols were coded
 * elsewhere (the GPU LF-group coder): lengths[] holds the code lengths of cluster c at offset
 * sum(alphabet[0..c)). */
typedef struct HydPrefixLayout {
    uint32_t rle_min_symbol, rle_min_length;
    const uint8_t *cluster_map;
    size_t num_dists, num_clusters;
    const HydUintConfig *config;
    const uint16_t *alphabet;
} HydPrefixLayout;
int hps_write_header(HydBits *out, const HydPrefixLayout *lay, const uint32_t *lengths, const char **err);
/* its fields in front of the alphabet sizes and codes — everything that does not depend on the symbols
 * (LZ77 parameters, cluster map, "prefix codes", hybrid-uint configurations) */
int hps_write_header_fixed(HydBits *out, const HydPrefixLayout *lay, const char **err);

/* pieces shared with the ANS stream header written by frame.c */
void hps_hybridize(uint32_t value, const HydUintConfig *cfg, HydSym *out);
int hps_write_cluster_map(const uint8_t *map, size_t num_dists, size_t num_clusters, HydBits *out, const char **err);
void hps_write_uint_config(HydBits *out, const HydUintConfig *cfg, int log_alphabet_size);

/* exposed for unit tests: depth-limited code lengths for `n` frequencies */
int hps_code_lengths(const uint32_t *freq, uint32_t *lengths, uint32_t n, int max_depth);

#endif /* HYD_PREFIX_H_ */
