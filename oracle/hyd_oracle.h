/*
 * oracle/hyd_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of hydrium's per-group encode hot path,
 * written from the algorithm, stage by stage, in a planar layout of our own.  It exists to check
 * the HIP kernels: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * it; the product library never links, imports or calls anything in oracle/.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_ref.py checks every stage below against the real
 * reference (oracle/_ref/libref_probe.so, built from /root/reference by oracle/Makefile) and
 * tests/test_golden.py checks it against the committed fixtures in tests/golden/ generated from
 * the reference by tests/golden/make_golden.py.
 *
 * Must be compiled with -ffp-contract=off on a target whose float evaluation is IEEE binary32
 * (x86-64 SSE): the reference's canonical output is the non-contracted one.
 */
#ifndef HYD_ORACLE_H_
#define HYD_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#define ORC_FMT_U8 0
#define ORC_FMT_U16 1
#define ORC_FMT_F32 2

#define ORC_MAX_ALPHABET 128 /* tokens are < 72 for 32-bit values with the (4,1,0) hybrid config */

#define ORC_OK 0
#define ORC_ERR_NAN (-14) /* same value as HYD_API_ERROR: "Invalid NaN Float" (format.c:123-126) */
#define ORC_ERR_NOMEM (-13)
#define ORC_ERR_INTERNAL (-15)

/* one emitted hybrid-uint symbol; field-for-field what the reference buffers (entropy.h:9-14) */
typedef struct OrcSymbol {
    uint16_t token;
    uint8_t cluster;
    uint8_t residue_bits;
    uint32_t residue;
} OrcSymbol;

/* Everything the hot path produces for one LF group (<= 2048x2048 px, <= 64 groups of 256x256). */
typedef struct OrcLfResult {
    size_t width, height;   /* pixels */
    size_t vbw, vbh;        /* 8x8 blocks across / down */
    size_t stride;          /* vbw * 8: row pitch of the planes below */
    size_t gcols, grows;    /* 256x256 groups across / down */
    size_t num_groups;

    /* planes are [3][vbh*8][stride], channel order X, Y, B */
    float *xyb;             /* after RGB->XYB + zero padding          (format.c:142-193) */
    float *dct;             /* after forward_dct, transposed per block (encoder.c:631-668) */
    int32_t *quant;         /* HF quantised ints at the same positions; DC slot holds 0 (encoder.c:783-823) */
    int32_t *dc;            /* [3][vbh][vbw] LF ints trunc(dc * {8192,1024,512}) (encoder.c:573,582) */
    uint8_t *nz;            /* [num_groups][1024][3] non-zero counts, row pitch = blocks per group row */

    OrcSymbol *symbols;     /* concatenated over groups in raster order (encoder.c:689-750) */
    size_t num_symbols;
    size_t *group_symbols;  /* [num_groups] */

    unsigned cluster_from, cluster_to;      /* clusters owned by this preset */
    uint16_t alphabet_size[256];            /* per global cluster id, only [from,to) are set */
    uint32_t freqs[256][ORC_MAX_ALPHABET];  /* normalised 12-bit frequencies (entropy.c:267-301) */
    unsigned max_alphabet_size;             /* running maximum after this LF group (entropy.c:459-460) */
    int log_alphabet_size;                  /* used for this LF group's tables (entropy.c:952,1073) */

    uint8_t *stream;        /* concatenated per-group HF sections, each zero-padded to a byte */
    size_t *group_offset;   /* [num_groups] byte offset into stream */
    size_t *group_bits;     /* [num_groups] exact bit length (preset id + ANS state + refills + residues) */
    size_t stream_bytes;
} OrcLfResult;

#ifdef __cplusplus
extern "C" {
#endif

/* ---- LUTs of the integer pixel path (format.c:58-83) ---- */
void orc_build_input_lut(uint16_t *lut, size_t size, int need_linearize);
void orc_build_bias_lut(float *lut /* 65536 entries */);

/* ---- scalar pieces exposed for unit tests ---- */
float orc_linearize(float x);                                   /* format.c:15-19 */
float orc_bias(float x);                                        /* format.c:21-31 */
void orc_hybridize(uint32_t value, OrcSymbol *out);             /* entropy.c:427-444 with config (4,1,0) */
uint32_t orc_pack_signed(int32_t v);                            /* math-functions.h:68-71 */
int orc_normalize_frequencies(uint32_t *freq, uint32_t alphabet_size);  /* entropy.c:267-301; returns 1 if unique */
/* 12-bit alias-table slot of (symbol, offset < freq[symbol]) (entropy.c:184-265,1102-1119). -1 on failure */
int orc_alias_slot(const uint32_t *freq, uint32_t alphabet_size, int log_alphabet_size, int unique,
                   uint32_t symbol, uint32_t offset);
/* HF context -> cluster map for num_presets presets (encoder.c:852-901); map has 1485*num_presets entries */
void orc_hf_cluster_map(uint8_t *map, unsigned num_presets);
/* forward DCT of one 8x8 block given as in[y][x]; out[row][col] laid out as the reference leaves it */
void orc_dct8x8(const float in[8][8], float out[8][8]);

/*
 * Run the whole hot path on one LF group.
 *   buf[3], row_stride, pixel_stride : exactly hyd_send_tile's buffer arguments (libhydrium.h:260-262),
 *                                      strides in samples, pointing at the LF group's first pixel
 *   width, height                    : LF group size in pixels (<= 2048)
 *   preset, num_presets              : this LF group's preset id and the frame's preset count
 *   max_alphabet_size                : in/out running maximum token+1 of the whole HF stream
 * Returns NULL and sets *err on failure.
 */
OrcLfResult *orc_encode_lf_group(const void *const buf[3], ptrdiff_t row_stride, ptrdiff_t pixel_stride, int fmt,
                                 int linear_light, size_t width, size_t height, unsigned preset,
                                 unsigned num_presets, unsigned *max_alphabet_size, int *err);
void orc_free_result(OrcLfResult *r);

/*
 * CPU-baseline driver ("port" leg of bench.py): the hot path over a whole interleaved RGB image,
 * LF groups in raster order, results discarded except the total section bytes and an FNV-1a
 * checksum over all group sections.  Single-threaded, like the reference.
 */
int orc_hot_path_image(const void *pixels, int fmt, size_t width, size_t height, int linear_light,
                       uint64_t *total_bytes, uint64_t *checksum);

#ifdef __cplusplus
}
#endif

#endif /* HYD_ORACLE_H_ */
