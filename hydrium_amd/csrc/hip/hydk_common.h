/*
 * hydk_common.h — shapes and device-side structs shared by the HIP kernels and their launcher.
 *
 * Vocabulary (JPEG XL / hydrium): an image is cut into LF groups of 2048x2048 px; an LF group into
 * <= 64 groups of 256x256 px; a group into <= 1024 varblocks of 8x8 px.  Each group's HF
 * coefficients are coded as one rANS chain into one "HF section" of the codestream.
 */
#ifndef HYDK_COMMON_H_
#define HYDK_COMMON_H_

#include <stdint.h>

#define HYDK_FMT_U8 0
#define HYDK_FMT_U16 1
#define HYDK_FMT_F32 2

#define HYDK_GROUPS_PER_LFG 64        /* 8 x 8 groups of 256 px in a 2048 px LF group */
#define HYDK_BLOCKS_PER_GROUP 1024    /* 32 x 32 varblocks */
#define HYDK_TOKENS_PER_GROUP 196608  /* hard maximum: 1024 blocks x 3 channels x (1 + 63) symbols */
#define HYDK_MAX_CLUSTERS 9           /* clusters one preset owns (scheme 0); fewer in the coarser schemes */
#define HYDK_ALPHABET 128             /* >= largest token + 1 (71 + 1) of the (4,1,0) hybrid-uint config */
#define HYDK_ANS_SLOTS 4096           /* 12-bit ANS precision */
#define HYDK_DC_PITCH 256             /* varblocks per row of an LF group's DC plane */
/* Token storage is sized for HYDK_DEFAULT_TOKEN_CAP records per group (1.5 symbols per pixel — photographic
 * content has 0.4 to 0.5, pure noise 2.9); a group that needs more raises HYDK_STATUS_OVERFLOW and the
 * host reruns the frame with the hard maximum. */
#define HYDK_DEFAULT_TOKEN_CAP 98304
/* Reversed per-group bit buffer of the wave-per-group entropy form, in 32-bit words, for `cap` token
 * records.  Worst case per symbol is one 16-bit refill plus a 30-bit residue; plus the 32-bit final
 * state and preset bits. */
#define HYDK_BITWORDS_FOR(cap) (((cap) * 46 + 64 + 31) / 32 + 1)
/* bits of the device status word */
#define HYDK_STATUS_BAD_SAMPLE 1u /* non-finite float sample */
#define HYDK_STATUS_TOKENS 2u     /* a group needed more token records than its array holds */
#define HYDK_STATUS_PAYLOAD 4u    /* the frame's sections need more bytes than the payload holds */
#define HYDK_STATUS_LAYOUT 8u     /* a float LF group met token arrays laid out for 4-byte records (host bug) */
#define HYDK_STATUS_INCONSISTENT 16u /* k_rans_emit: a section's records and refill flags hold more bits than its chain counted (a bug upstream; the frame fails) */
#define HYDK_STATUS_OVERFLOW (HYDK_STATUS_TOKENS | HYDK_STATUS_PAYLOAD | HYDK_STATUS_LAYOUT) /* later stages skip the frame */

/* Token record (8 bytes) written by the transform kernel and read by the rANS kernel:
 *   lo: bits 0-7 token, 8-11 preset-local cluster, 16-21 residue bit count;  hi: residue bits. */
#define HYDK_REC_LO(token, cluster, rbits) ((uint32_t)(token) | ((uint32_t)(cluster) << 8) | ((uint32_t)(rbits) << 16))
/* Integer sample formats bound every quantised coefficient below 2^13 (K1), so their record fits 4 bytes:
 *   bits 0-3 residue bit count (at most 12), 4-14 symbol = preset-local cluster * HYDK_REC32_TOKENS + token (what the
 *   transform kernel's histogram is indexed by; `record & 0x7FF0` IS the byte offset of the symbol's 16-byte row in the
 *   lane-form chain's operand table: one instruction, where a shift and a mask were two of a step's sixteen), 16-31 residue bits.
 * HYDK_REC32_TO_LO turns one back into the lo word of the 8-byte form (symbol / 40 by multiplication: exact below 2^11,
 * asserted in kernels.hip). */
#define HYDK_REC32_TOKENS 40u /* every token of an integer frame is below 36 */
#define HYDK_REC32(symbol, rbits, residue) ((uint32_t)(rbits) | ((uint32_t)(symbol) << 4) | ((uint32_t)(residue) << 16))
#define HYDK_REC32_SYMBOL(r) (((r) >> 4) & 0x7FFu)
#define HYDK_REC32_RBITS(r) ((r) & 0xFu)
#define HYDK_REC32_CLUSTER(r) ((HYDK_REC32_SYMBOL(r) * 1639u) >> 16)
#define HYDK_REC32_TO_LO(r) \
    ((HYDK_REC32_SYMBOL(r) - HYDK_REC32_CLUSTER(r) * HYDK_REC32_TOKENS) | (HYDK_REC32_CLUSTER(r) << 8) | (HYDK_REC32_RBITS(r) << 16))

/* Per-LF-group ANS coding tables produced by the table kernel, consumed by the rANS kernel. */
typedef struct HydkTables {
    uint32_t freq[HYDK_MAX_CLUSTERS][HYDK_ALPHABET];   /* normalised 12-bit frequencies (also read back by the host) */
    uint32_t fb[HYDK_MAX_CLUSTERS][HYDK_ALPHABET];     /* freq | (cumulative base << 16) */
    uint32_t magic[HYDK_MAX_CLUSTERS][HYDK_ALPHABET];  /* floor(2^32 / freq) (0xFFFFFFFF for freq 1) */
    /* (symbol, remainder) -> alias-table slot.  Symbol s owns entries [2*base, 2*base + 2*freq):
     * the first freq hold slot(r); the second freq hold slot(r - freq) + 4096, so that the
     * one-too-small quotient of the multiply-high division is repaired by the same lookup. */
    uint16_t inv[HYDK_MAX_CLUSTERS][2 * HYDK_ANS_SLOTS];
    /* the plain form, slot(r) at base + r for r < freq: half the LDS, used by the kernel variant that
     * packs a whole LF group into one workgroup and repairs the quotient with two extra operations */
    uint16_t inv1[HYDK_MAX_CLUSTERS][HYDK_ANS_SLOTS];
    uint32_t alphabet[HYDK_MAX_CLUSTERS];              /* largest token + 1 seen per cluster */
    uint32_t log_alphabet_size;                        /* max(5, ceil log2 of the running max alphabet) */
    uint32_t running_max_alphabet;                     /* after this LF group, in send order */
    uint32_t error;                                    /* non-zero: table construction failed */
    uint32_t pad;
} HydkTables;

/* Everything the kernels need to know about one LF group (one entry per slot, in send order). */
typedef struct HydkLfJob {
    const void *src[3];      /* R, G, B sample pointers of the LF group's first pixel (device memory) */
    long long row_stride;    /* in samples */
    long long pixel_stride;  /* in samples */
    int fmt;                 /* HYDK_FMT_* */
    int linear_light;
    int width, height;       /* LF group size in pixels */
    int gcols, grows;        /* groups across / down */
    int scheme;              /* HF clustering scheme 0..3 = 9 / 3 / 2 / 1 clusters per preset */
    int use_luts;            /* 1: gather from the uploaded LUTs instead of evaluating them in registers */
    unsigned preset;         /* HF preset id of this LF group (written in front of each group section) */
    int pad0;
    const uint16_t *in_lut8;   /* 256 entries   */
    const uint16_t *in_lut16;  /* 65536 entries */
    const float *bias_lut;     /* 65536 entries */
    void *tokens;            /* [groups][tok_cap] records of 8 bytes (float input) or 4 (integer input); group pitch tok_cap * rec_bytes */
    uint32_t tok_cap;        /* records a group's token array can hold */
    uint32_t rec_bytes;      /* record size the token array was laid out for: 4 (integer input only) or 8 */
    uint32_t *sym_count;     /* [groups] */
    uint32_t *rbits_total;   /* [groups] residue bits of the group's symbols, summed */
    uint32_t *hist;          /* [HYDK_MAX_CLUSTERS][HYDK_ALPHABET], zeroed before launch */
    uint32_t *alpha_max;     /* [1] largest token + 1 over this LF group's symbols, zeroed before launch */
    int32_t *dc;             /* [3][HYDK_DC_PITCH][HYDK_DC_PITCH] LF ints */
    float *dbg_xyb;          /* optional [3][2048][2048] dumps for parity tests, else NULL */
    float *dbg_dct;
    int32_t *dbg_quant;
} HydkLfJob;

/* ---- LF-group coder (the modular sub-stream of the LF coefficients, encoder.c:560-596) ----
 * One LF group sends 3 x vbw x vbh residuals through a prefix-coded stream with hybrid-uint
 * config (7,1,1) and LZ77 used as run-length coding (min symbol 16384, min length 3).  Only these
 * tokens can occur: literals 0..227 and run tokens 16385..16508; the coder keeps them in a compact
 * index space of HYDK_LF_CODES entries: [0,256) literals, [256,384) token 16384 + (i - 256). */
#define HYDK_LF_SYMBOLS (3 * HYDK_DC_PITCH * HYDK_DC_PITCH)
#define HYDK_LF_CODES 384
#define HYDK_LF_RUN_BASE 16384
/* worst case per coefficient: 15-bit code + 29 residue bits + a 15-bit run code = 59 bits */
#define HYDK_LF_BITWORDS (HYDK_LF_SYMBOLS * 2 + 2)

typedef struct HydkLfStream {
    uint32_t bit_count;                /* bits of symbol data in the slot's bit buffer */
    uint32_t alphabet;                 /* largest token + 1 of the value cluster */
    uint32_t run_pairs;                /* (run token, distance) pairs sent; the distance cluster only ever sees token 1 */
    uint32_t error;                    /* non-zero: code construction failed */
    uint32_t offset;                   /* byte offset of the symbol data in the frame's packed LF payload */
    uint32_t pad[3];
    uint8_t lengths[HYDK_LF_CODES];    /* prefix-code length per compact token index (0 = unused) */
} HydkLfStream;

#endif /* HYDK_COMMON_H_ */
