/*
 * oracle/hyd_oracle.c — TEST INFRASTRUCTURE ONLY (see hyd_oracle.h).
 *
 * CPU restatement of the reference's per-group hot path.  Each function cites the reference lines
 * (relative to /root/reference/src/libhydrium/) whose arithmetic it restates.  Layout and control
 * structure are ours (planar float/int planes, one explicit stage per function, tables derived
 * from their defining formulas where one exists); the arithmetic — operation order, float
 * rounding points, integer widths — is the reference's, because the bytes must match.
 */
#include "hyd_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * small helpers
 * ---------------------------------------------------------------------------------------- */

static int floor_log2_u32(uint32_t v) { return 31 - __builtin_clz(v); }                   /* math-functions.h:8-17 */
static int ceil_log2_u32(uint32_t v) { return floor_log2_u32(v) + ((v & (v - 1)) != 0); } /* math-functions.h:66 */

uint32_t orc_pack_signed(int32_t v) {
    const uint32_t w = (uint32_t)v;
    return (w << 1) ^ (0u - (w >> 31));
}

/* ------------------------------------------------------------------------------------------
 * constant tables
 * ---------------------------------------------------------------------------------------- */

/* The seven distinct |cos| magnitudes of the scaled 8-point DCT-II, as the decimal literals the
 * reference spells out (encoder.c:32-40); index t is the folded angle t*pi/16. They are double
 * literals narrowed to float at compile time, as in the reference's initialiser. */
static const float dct_mag[8] = {0.0f, 0.17338, 0.16332, 0.146984, 0.125, 0.0982119, 0.0676495, 0.0344874};

static float dct_coef[7][8];   /* [k-1][n] = cos((2n+1) k pi/16) * sqrt(2)/8 */
static uint8_t zz_row[64], zz_col[64]; /* position of zig-zag coefficient j inside the stored block */
static uint8_t freq_ctx[64];   /* encoder.c:53-58 */
static uint16_t nnz_ctx[64];   /* encoder.c:60-66 */
static int tables_ready;

/* HF quantisation weights per channel (X, Y, B) in zig-zag order (encoder.c:74-93). */
static const int32_t quant_weight[3][64] = {
    {1969, 1969, 1969, 1962, 1969, 1962, 1655, 1885, 1885, 1655, 1397, 1610, 1704, 1610, 1397, 1178,
     1368, 1494, 1494, 1368, 1178, 994,  1159, 1289, 1340, 1289, 1159, 994,  839,  980,  1104, 1178,
     1178, 1104, 980,  839,  829,  941,  1023, 1054, 1023, 941,  829,  800,  881,  928,  928,  881,
     800,  755,  809,  829,  809,  755,  663,  731,  731,  663,  491,  524,  491,  349,  349,  239},
    {280, 280, 280, 279, 280, 279, 245, 271, 271, 245, 214, 239, 250, 239, 214, 188,
     211, 226, 226, 211, 188, 164, 185, 201, 207, 201, 185, 164, 144, 163, 178, 188,
     188, 178, 163, 144, 143, 157, 168, 172, 168, 157, 143, 139, 150, 156, 156, 150,
     139, 133, 140, 143, 140, 133, 125, 129, 129, 125, 116, 118, 116, 107, 107, 98},
    {256, 147, 147, 85, 117, 85, 60, 78, 78, 60, 43, 56, 63, 56, 43, 43,
     43,  48,  48,  43, 43,  42, 43, 43, 43, 43, 43, 42, 29, 41, 43, 43,
     43,  43,  41,  29, 29,  37, 43, 43, 43, 37, 29, 27, 33, 36, 36, 33,
     27,  24,  27,  29, 27,  24, 20, 22, 22, 20, 15, 16, 15, 10, 10, 7},
};
static const float hf_mult_f = 5.0f;                           /* encoder.c:95 */
static const float lf_shift[3] = {8192.f, 1024.f, 512.f};      /* encoder.c:573 */

static void init_tables(void) {
    if (tables_ready)
        return;
    /* cosine table from the angle folding; signs follow cos(m*pi/16), m = (2n+1)k mod 32 */
    for (int k = 1; k < 8; k++) {
        for (int n = 0; n < 8; n++) {
            int m = ((2 * n + 1) * k) & 31;
            if (m > 16)
                m = 32 - m;
            float sign = 1.0f;
            if (m > 8) {
                m = 16 - m;
                sign = -1.0f;
            }
            dct_coef[k - 1][n] = sign * dct_mag[m];
        }
    }
    /* zig-zag: anti-diagonals d = col+row; odd d walks col downwards, even d upwards.  The
     * reference table holds {x, y} with x added to the column and y to the row
     * (encoder.c:42-51,805-806). */
    int j = 0;
    for (int d = 0; d < 15; d++) {
        if (d & 1) {
            for (int col = d < 8 ? d : 7; col >= 0 && d - col < 8; col--, j++) {
                zz_col[j] = (uint8_t)col;
                zz_row[j] = (uint8_t)(d - col);
            }
        } else {
            for (int col = d < 8 ? 0 : d - 7; col < 8 && col <= d; col++, j++) {
                zz_col[j] = (uint8_t)col;
                zz_row[j] = (uint8_t)(d - col);
            }
        }
    }
    /* frequency context: 0,0, then 1..14 singly, 15..22 in pairs, 23..30 in fours */
    freq_ctx[0] = freq_ctx[1] = 0;
    for (int i = 2; i < 16; i++)
        freq_ctx[i] = (uint8_t)(i - 1);
    for (int i = 16; i < 32; i++)
        freq_ctx[i] = (uint8_t)(15 + (i - 16) / 2);
    for (int i = 32; i < 64; i++)
        freq_ctx[i] = (uint8_t)(23 + (i - 32) / 4);
    /* non-zero-count context offsets */
    for (int i = 0; i < 64; i++) {
        nnz_ctx[i] = i < 2 ? 0 : i < 3 ? 31 : i < 5 ? 62 : i < 9 ? 93 : i < 13 ? 123 : i < 21 ? 152 : i < 33 ? 180 : 206;
    }
    tables_ready = 1;
}

/* ------------------------------------------------------------------------------------------
 * stage 1: RGB -> XYB
 * ---------------------------------------------------------------------------------------- */

float orc_linearize(float x) {                                              /* format.c:15-19 */
    if (x <= 0.0404482362771082f)
        return 0.07739938080495357f * x;
    return 0.003094300919832f + x * (-0.009982599f + x * (0.72007737769f + 0.2852804880f * x));
}

static float cube_root(float x) {                                           /* format.c:21-27 */
    union { float f; uint32_t u; } z;
    z.f = x;
    z.u = 0x548c39cbu - z.u / 3u;
    z.f *= 1.5015480449f - 0.534850249f * x * z.f * z.f * z.f;
    z.f *= 1.333333985f - 0.33333333f * x * z.f * z.f * z.f;
    return 1.0f / z.f;
}

float orc_bias(float x) { return cube_root(x + 0.0037930732552754493f) - 0.155954f; }   /* format.c:29-31 */

static uint16_t float_to_u16(float x) {                                     /* format.c:33-36 */
    const int32_t y = (int32_t)(x * 65535.f + 0.5f);
    return (uint16_t)(y < 0 ? 0 : y > 65535 ? 65535 : y);
}

void orc_build_input_lut(uint16_t *lut, size_t size, int need_linearize) {  /* format.c:58-71 */
    const float factor = 1.0f / (size - 1.0f);
    for (size_t i = 0; i < size; i++) {
        const float f = i * factor;
        lut[i] = float_to_u16(need_linearize ? orc_linearize(f) : f);
    }
}

void orc_build_bias_lut(float *lut) {                                       /* format.c:73-83 */
    const float factor = 1.0f / (65536 - 1.0f);
    for (size_t i = 0; i < 65536; i++)
        lut[i] = orc_bias(i * factor);
}

static int float_is_finite(float x) {                                       /* math-functions.h:73-76 */
    union { float f; uint32_t u; } z;
    z.f = x;
    return (z.u & 0x7f800000u) != 0x7f800000u;
}

/* L,M,S gamma values -> X,Y,B (format.c:42-45,52-55) */
static void lms_to_xyb(float l, float m, float s, float *X, float *Y, float *B) {
    const float y = (l + m) * 0.5f;
    *Y = y;
    *X = y - m;
    *B = s - y;
}

static int stage_xyb(const void *const buf[3], ptrdiff_t row_stride, ptrdiff_t pixel_stride, int fmt,
                     int linear_light, size_t width, size_t height, size_t stride, size_t rows, float *planes) {
    const size_t plane = stride * rows;
    float *PX = planes, *PY = planes + plane, *PB = planes + 2 * plane;
    memset(planes, 0, 3 * plane * sizeof(float));   /* edge padding is XYB = 0 (format.c:182-191) */

    if (fmt == ORC_FMT_U8 || fmt == ORC_FMT_U16) {
        const size_t lut_size = fmt == ORC_FMT_U8 ? 256 : 65536;
        uint16_t *in_lut = malloc(lut_size * sizeof(uint16_t));
        float *bias_lut = malloc(65536 * sizeof(float));
        if (!in_lut || !bias_lut) {
            free(in_lut);
            free(bias_lut);
            return ORC_ERR_NOMEM;
        }
        orc_build_input_lut(in_lut, lut_size, !linear_light);
        orc_build_bias_lut(bias_lut);
        for (size_t y = 0; y < height; y++) {
            for (size_t x = 0; x < width; x++) {
                const ptrdiff_t off = (ptrdiff_t)y * row_stride + (ptrdiff_t)x * pixel_stride;
                uint32_t r, g, b;
                if (fmt == ORC_FMT_U8) {
                    r = in_lut[((const uint8_t *)buf[0])[off]];
                    g = in_lut[((const uint8_t *)buf[1])[off]];
                    b = in_lut[((const uint8_t *)buf[2])[off]];
                } else {
                    r = in_lut[((const uint16_t *)buf[0])[off]];
                    g = in_lut[((const uint16_t *)buf[1])[off]];
                    b = in_lut[((const uint16_t *)buf[2])[off]];
                }
                /* format.c:49-51: 16.16 fixed-point LMS mix in 32-bit unsigned, high half indexes the LUT */
                const float l = bias_lut[((19661u * r + 40761u * g + 5112u * b) >> 16) & 0xFFFFu];
                const float m = bias_lut[((15073u * r + 45350u * g + 5112u * b) >> 16) & 0xFFFFu];
                const float s = bias_lut[((15953u * r + 13419u * g + 36163u * b) >> 16) & 0xFFFFu];
                const size_t p = y * stride + x;
                lms_to_xyb(l, m, s, &PX[p], &PY[p], &PB[p]);
            }
        }
        free(in_lut);
        free(bias_lut);
        return ORC_OK;
    }

    /* float path (format.c:111-140, 38-46) */
    for (size_t y = 0; y < height; y++) {
        for (size_t x = 0; x < width; x++) {
            const ptrdiff_t off = (ptrdiff_t)y * row_stride + (ptrdiff_t)x * pixel_stride;
            float r = ((const float *)buf[0])[off];
            float g = ((const float *)buf[1])[off];
            float b = ((const float *)buf[2])[off];
            if (!float_is_finite(r) || !float_is_finite(g) || !float_is_finite(b))
                return ORC_ERR_NAN;
            if (!linear_light) {
                r = orc_linearize(r);
                g = orc_linearize(g);
                b = orc_linearize(b);
            }
            const float l = orc_bias(0.3f * r + 0.622f * g + 0.078f * b);
            const float m = orc_bias(0.23f * r + 0.692f * g + 0.078f * b);
            const float s = orc_bias(0.243423f * r + 0.204767f * g + 0.55181f * b);
            const size_t p = y * stride + x;
            lms_to_xyb(l, m, s, &PX[p], &PY[p], &PB[p]);
        }
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------------------------------
 * stage 2: 8x8 forward DCT (encoder.c:631-668)
 * ---------------------------------------------------------------------------------------- */

/* one 8-point transform in the reference's summation order: DC is the left-to-right sum times
 * 0.125; each AC term accumulates x[n]*c[k][n] for n = 0..7 starting from +0.0 */
static void dct8(const float x[8], float out[8]) {
    float dc = x[0];
    for (int n = 1; n < 8; n++)
        dc += x[n];
    out[0] = dc * 0.125f;
    for (int k = 1; k < 8; k++) {
        float acc = 0.0f;
        for (int n = 0; n < 8; n++)
            acc += x[n] * dct_coef[k - 1][n];
        out[k] = acc;
    }
}

void orc_dct8x8(const float in[8][8], float out[8][8]) {
    init_tables();
    float rowpass[8][8]; /* [y][horizontal frequency] */
    for (int y = 0; y < 8; y++)
        dct8(in[y], rowpass[y]);
    for (int h = 0; h < 8; h++) {
        float col[8], v[8];
        for (int y = 0; y < 8; y++)
            col[y] = rowpass[y][h];
        dct8(col, v);
        /* the reference stores pass2[x][y] at row y, col x (encoder.c:660-664): the block is left
         * transposed, i.e. row index = horizontal frequency, column index = vertical frequency */
        for (int k = 0; k < 8; k++)
            out[h][k] = v[k];
    }
}

static void stage_dct(const float *xyb, float *dct, size_t vbw, size_t vbh, size_t stride) {
    const size_t plane = stride * vbh * 8;
    for (int c = 0; c < 3; c++) {
        const float *src = xyb + c * plane;
        float *dst = dct + c * plane;
        for (size_t by = 0; by < vbh; by++) {
            for (size_t bx = 0; bx < vbw; bx++) {
                float in[8][8], out[8][8];
                for (int y = 0; y < 8; y++)
                    memcpy(in[y], src + (by * 8 + y) * stride + bx * 8, 8 * sizeof(float));
                orc_dct8x8(in, out);
                for (int y = 0; y < 8; y++)
                    memcpy(dst + (by * 8 + y) * stride + bx * 8, out[y], 8 * sizeof(float));
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * stage 3: quantisation (HF: encoder.c:783-823, LF: encoder.c:573,582)
 * ---------------------------------------------------------------------------------------- */

typedef struct GroupGeom {
    size_t px, py;   /* pixel origin inside the LF group */
    size_t gbw, gbh; /* blocks across / down */
} GroupGeom;

static GroupGeom group_geom(const OrcLfResult *r, size_t g) {
    GroupGeom gg;
    const size_t gy = g / r->gcols, gx = g % r->gcols;
    gg.px = gx << 8;
    gg.py = gy << 8;
    const size_t gw = gg.px + 256 > r->width ? r->width - gg.px : 256;
    const size_t gh = gg.py + 256 > r->height ? r->height - gg.py : 256;
    gg.gbw = (gw + 7) >> 3;
    gg.gbh = (gh + 7) >> 3;
    return gg;
}

static void stage_quant(OrcLfResult *r) {
    const size_t plane = r->stride * r->vbh * 8;
    memset(r->quant, 0, 3 * plane * sizeof(int32_t));
    memset(r->nz, 0, r->num_groups * 1024 * 3);
    for (size_t g = 0; g < r->num_groups; g++) {
        const GroupGeom gg = group_geom(r, g);
        for (size_t by = 0; by < gg.gbh; by++) {
            for (size_t bx = 0; bx < gg.gbw; bx++) {
                for (int c = 0; c < 3; c++) {
                    const float *src = r->dct + c * plane;
                    int32_t *dst = r->quant + c * plane;
                    unsigned count = 0;
                    for (int j = 1; j < 64; j++) {
                        const size_t p = (gg.py + by * 8 + zz_row[j]) * r->stride + gg.px + bx * 8 + zz_col[j];
                        /* (coef * weight) * 5, then C truncation; +-1 fall into the dead zone */
                        const int32_t q = (int32_t)(src[p] * quant_weight[c][j] * hf_mult_f);
                        if (q > 1 || q < -1) {
                            dst[p] = q;
                            count++;
                        }
                    }
                    r->nz[(g * 1024 + by * gg.gbw + bx) * 3 + c] = (uint8_t)count;
                }
            }
        }
    }
    /* LF ints: truncation of dc * per-channel shift */
    for (int c = 0; c < 3; c++) {
        for (size_t by = 0; by < r->vbh; by++)
            for (size_t bx = 0; bx < r->vbw; bx++)
                r->dc[(c * r->vbh + by) * r->vbw + bx] =
                    (int32_t)(r->dct[c * plane + by * 8 * r->stride + bx * 8] * lf_shift[c]);
    }
}

/* ------------------------------------------------------------------------------------------
 * stage 4: tokenisation (encoder.c:670-750, entropy.c:427-471)
 * ---------------------------------------------------------------------------------------- */

void orc_hybridize(uint32_t value, OrcSymbol *out) {
    /* config split_exponent 4, msb_in_token 1, lsb_in_token 0 (encoder.c:908) */
    if (value < 16) {
        out->token = (uint16_t)value;
        out->residue = 0;
        out->residue_bits = 0;
        return;
    }
    const int top = floor_log2_u32(value);
    const int n = top - 1;                          /* bits below the one explicit MSB-side bit */
    out->residue = value & ((1u << n) - 1u);
    out->residue_bits = (uint8_t)n;
    out->token = (uint16_t)(16 + (((uint32_t)(n - 4 + 1)) << 1 | ((value >> n) & 1u)));
}

void orc_hf_cluster_map(uint8_t *map, unsigned num_presets) {
    for (unsigned p = 0; p < num_presets; p++) {
        uint8_t *m = map + 1485u * p;
        for (unsigned ctx = 0; ctx < 1485; ctx++) {
            const int is_coef = ctx >= 111;
            if (num_presets * 9 <= 256)
                m[ctx] = (uint8_t)(9 * p + (is_coef ? 3 + (ctx - 111) % 6 : ctx % 3));
            else if (num_presets * 3 <= 256)
                m[ctx] = (uint8_t)(3 * p + (is_coef ? 1 + ((ctx - 111) & 1) : 0));
            else if (num_presets * 2 <= 256)
                m[ctx] = (uint8_t)(2 * p + is_coef);
            else
                m[ctx] = (uint8_t)p;
        }
    }
}

static unsigned predicted_nz(const uint8_t *nz, size_t by, size_t bx, size_t gbw, int c) {  /* encoder.c:670-678 */
    if (!bx && !by)
        return 32;
    if (!bx)
        return nz[((by - 1) * gbw) * 3 + c];
    if (!by)
        return nz[(bx - 1) * 3 + c];
    return (nz[((by - 1) * gbw + bx) * 3 + c] + (unsigned)nz[(by * gbw + bx - 1) * 3 + c] + 1) >> 1;
}

static unsigned nz_context(unsigned predicted) {                                             /* encoder.c:680-687 */
    if (predicted < 8)
        return predicted;
    if (predicted > 64)
        predicted = 64;
    return 4 + (predicted >> 1);
}

typedef struct SymVec {
    OrcSymbol *v;
    size_t n, cap;
} SymVec;

static int push_symbol(SymVec *sv, const uint8_t *cmap, size_t ctx, uint32_t value, OrcLfResult *r) {
    if (sv->n == sv->cap) {
        size_t ncap = sv->cap ? sv->cap * 2 : 1 << 16;
        OrcSymbol *nv = realloc(sv->v, ncap * sizeof(OrcSymbol));
        if (!nv)
            return ORC_ERR_NOMEM;
        sv->v = nv;
        sv->cap = ncap;
    }
    OrcSymbol *s = &sv->v[sv->n++];
    orc_hybridize(value, s);
    s->cluster = cmap[ctx];
    if (s->token + 1u > r->alphabet_size[s->cluster])
        r->alphabet_size[s->cluster] = (uint16_t)(s->token + 1u);
    if (s->token + 1u > r->max_alphabet_size)
        r->max_alphabet_size = s->token + 1u;
    return ORC_OK;
}

static int stage_tokenize(OrcLfResult *r, unsigned preset, const uint8_t *cmap) {
    const size_t plane = r->stride * r->vbh * 8;
    SymVec sv = {0};
    for (size_t g = 0; g < r->num_groups; g++) {
        const GroupGeom gg = group_geom(r, g);
        const uint8_t *nz = r->nz + g * 1024 * 3;
        const size_t before = sv.n;
        for (size_t by = 0; by < gg.gbh; by++) {
            for (size_t bx = 0; bx < gg.gbw; bx++) {
                for (unsigned visit = 0; visit < 3; visit++) {
                    const int c = visit < 2 ? 1 - (int)visit : 2;      /* visit order Y, X, B (encoder.c:712) */
                    const int32_t *q = r->quant + c * plane;
                    const size_t origin = (gg.py + by * 8) * r->stride + gg.px + bx * 8;
                    unsigned remaining = nz[(by * gg.gbw + bx) * 3 + c];
                    const size_t ctx_nz = 1485u * preset + 3 * nz_context(predicted_nz(nz, by, bx, gg.gbw, c)) + visit;
                    int ret = push_symbol(&sv, cmap, ctx_nz, remaining, r);
                    if (ret)
                        goto fail;
                    if (!remaining)
                        continue;
                    const size_t base = 1485u * preset + 458u * visit + 111u;
                    unsigned prev_nonzero = remaining <= 4;             /* encoder.c:730, k == 0 case */
                    for (int j = 1; j < 64; j++) {
                        const int32_t coef = q[origin + zz_row[j] * r->stride + zz_col[j]];
                        const size_t ctx = base + prev_nonzero + 2u * (nnz_ctx[remaining] + freq_ctx[j]);
                        ret = push_symbol(&sv, cmap, ctx, orc_pack_signed(coef), r);
                        if (ret)
                            goto fail;
                        prev_nonzero = coef != 0;
                        if (coef && !--remaining)
                            break;
                    }
                }
            }
        }
        r->group_symbols[g] = sv.n - before;
    }
    r->symbols = sv.v;
    r->num_symbols = sv.n;
    return ORC_OK;
fail:
    free(sv.v);
    return ORC_ERR_NOMEM;
}

/* ------------------------------------------------------------------------------------------
 * stage 5: histograms -> 12-bit ANS frequencies -> alias table (entropy.c:184-301,526-544,943-978)
 * ---------------------------------------------------------------------------------------- */

int orc_normalize_frequencies(uint32_t *freq, uint32_t alphabet_size) {
    uint64_t total = 0;
    for (uint32_t k = 0; k < alphabet_size; k++)
        total += freq[k];
    if (!total)
        return -1;
    uint64_t scaled_total = 0;
    for (uint32_t k = 0; k < alphabet_size; k++) {
        if (!freq[k])
            continue;
        freq[k] = (uint32_t)((((uint64_t)freq[k] << 12) / total) & 0xFFFFu);
        if (!freq[k])
            freq[k] = 1;
        scaled_total += freq[k];
    }
    /* shave the excess off the tail, never below 1 */
    size_t j = alphabet_size - 1;
    while (scaled_total > 4096) {
        const uint64_t excess = scaled_total - 4096;
        if (excess < freq[j]) {
            freq[j] -= (uint32_t)excess;
            scaled_total -= excess;
            break;
        } else if (freq[j] > 1) {
            scaled_total -= freq[j] - 1;
            freq[j] = 1;
        }
        j--;
    }
    /* any shortfall goes to token 0, used or not */
    freq[0] += (uint32_t)(4096 - scaled_total);
    return freq[alphabet_size - 1] == 4096;
}

typedef struct AliasTable {
    int log_bucket;
    uint32_t table_size;
    uint32_t cutoff[256]; /* slots [0,cutoff) of bucket i keep symbol i with offset = position */
    uint32_t other[256];  /* symbol owning the rest of bucket i */
    uint32_t shift[256];  /* offset = shift + position for that other symbol (mod 2^32) */
} AliasTable;

static int build_alias(AliasTable *t, const uint32_t *freq, uint32_t alphabet_size, int log_alphabet_size, int unique) {
    memset(t, 0, sizeof(*t));
    t->log_bucket = 12 - log_alphabet_size;
    t->table_size = 1u << log_alphabet_size;
    const uint32_t bucket = 1u << t->log_bucket;
    if (unique) {
        /* a lone symbol owns every slot: slot number == offset (entropy.c:195-200) */
        for (uint32_t i = 0; i < t->table_size; i++) {
            t->other[i] = alphabet_size - 1;
            t->shift[i] = i * bucket;
        }
        return ORC_OK;
    }
    uint8_t under[256], over[256];
    size_t n_under = 0, n_over = 0;
    for (uint32_t s = 0; s < alphabet_size; s++) {
        t->cutoff[s] = freq[s];
        if (freq[s] < bucket)
            under[n_under++] = (uint8_t)s;
        else if (freq[s] > bucket)
            over[n_over++] = (uint8_t)s;
    }
    for (uint32_t s = alphabet_size; s < t->table_size; s++)
        under[n_under++] = (uint8_t)s;
    /* LIFO pairing of an under-full bucket with an over-full symbol (entropy.c:217-231) */
    while (n_over) {
        if (!n_under)
            return ORC_ERR_INTERNAL;
        const uint8_t u = under[--n_under];
        const uint8_t o = over[--n_over];
        const uint32_t moved = bucket - t->cutoff[u];
        t->cutoff[o] -= moved;
        t->shift[u] = t->cutoff[o];
        t->other[u] = o;
        if (t->cutoff[o] < bucket)
            under[n_under++] = o;
        else if (t->cutoff[o] > bucket)
            over[n_over++] = o;
    }
    for (uint32_t i = 0; i < t->table_size; i++) {
        if (t->cutoff[i] == bucket) {
            t->other[i] = i;
            t->cutoff[i] = t->shift[i] = 0;
        } else {
            t->shift[i] -= t->cutoff[i];
        }
    }
    return ORC_OK;
}

/* (symbol, offset) -> slot. The reference walks a per-symbol list (own bucket first, then the
 * buckets it was spilled into in ascending order, entropy.c:1104-1113); the mapping is a
 * bijection so the first hit in the same order is the only hit. */
static int alias_slot(const AliasTable *t, uint32_t symbol, uint32_t offset) {
    const uint32_t bucket = 1u << t->log_bucket;
    if (offset < t->cutoff[symbol] && offset < bucket)
        return (int)((symbol << t->log_bucket) | offset);
    for (uint32_t i = 0; i < t->table_size; i++) {
        if (t->other[i] != symbol)
            continue;
        const uint32_t pos = offset - t->shift[i];
        if (pos < bucket && pos >= t->cutoff[i])
            return (int)((i << t->log_bucket) | pos);
    }
    return -1;
}

int orc_alias_slot(const uint32_t *freq, uint32_t alphabet_size, int log_alphabet_size, int unique,
                   uint32_t symbol, uint32_t offset) {
    AliasTable t;
    if (build_alias(&t, freq, alphabet_size, log_alphabet_size, unique))
        return -1;
    return alias_slot(&t, symbol, offset);
}

/* ------------------------------------------------------------------------------------------
 * stage 6: reverse rANS + forward bit emission (entropy.c:1064-1159, bitwriter.c:110-124)
 * ---------------------------------------------------------------------------------------- */

typedef struct BitBuf {
    uint8_t *data;
    size_t bits, cap_bytes;
} BitBuf;

static int put_bits(BitBuf *b, uint64_t value, int nbits) { /* LSB-first, like hyd_write */
    if (nbits <= 0)
        return ORC_OK;
    const size_t need = (b->bits + (size_t)nbits + 7) / 8 + 8;
    if (need > b->cap_bytes) {
        size_t ncap = b->cap_bytes ? b->cap_bytes : 4096;
        while (ncap < need)
            ncap *= 2;
        uint8_t *nd = realloc(b->data, ncap);
        if (!nd)
            return ORC_ERR_NOMEM;
        memset(nd + b->cap_bytes, 0, ncap - b->cap_bytes);
        b->data = nd;
        b->cap_bytes = ncap;
    }
    if (nbits < 64)
        value &= (UINT64_C(1) << nbits) - 1;
    /* callers never pass more than 32 bits at a time, so value << 7 still fits in 64 bits */
    uint64_t shifted = value << (b->bits & 7);
    for (size_t k = b->bits >> 3; shifted; k++, shifted >>= 8)
        b->data[k] |= (uint8_t)shifted;
    b->bits += (size_t)nbits;
    return ORC_OK;
}

/* Encode one group's symbols. refill[p] != 0 marks a 16-bit word emitted just before residue p. */
static int rans_encode_group(const OrcSymbol *sym, size_t n, uint32_t (*freqs)[ORC_MAX_ALPHABET],
                             const AliasTable *alias, unsigned preset, int preset_bits, BitBuf *out) {
    int ret = put_bits(out, preset, preset_bits);   /* encoder.c:945 */
    if (ret || !n)
        return ret;                                  /* a group with no symbols writes nothing more */
    uint8_t *has_refill = calloc(n, 1);
    uint16_t *refill = malloc(n * sizeof(uint16_t));
    if (!has_refill || !refill) {
        free(has_refill);
        free(refill);
        return ORC_ERR_NOMEM;
    }
    uint32_t state = 0x130000u;
    for (size_t p = n; p-- > 0;) {
        const uint32_t f = freqs[sym[p].cluster][sym[p].token];
        if ((state >> 20) >= f) {
            has_refill[p] = 1;
            refill[p] = (uint16_t)(state & 0xFFFF);
            state >>= 16;
        }
        const uint32_t q = state / f;
        const int slot = alias_slot(&alias[sym[p].cluster], sym[p].token, state - q * f);
        if (slot < 0) {
            free(has_refill);
            free(refill);
            return ORC_ERR_INTERNAL;
        }
        state = (q << 12) | (uint32_t)slot;
    }
    /* final state: low half first, then high half (entropy.c:1127-1130 popped LIFO) */
    ret = put_bits(out, state & 0xFFFF, 16);
    if (!ret)
        ret = put_bits(out, state >> 16, 16);
    for (size_t p = 0; p < n && !ret; p++) {
        if (has_refill[p])
            ret = put_bits(out, refill[p], 16);
        if (!ret)
            ret = put_bits(out, sym[p].residue, sym[p].residue_bits);
    }
    free(has_refill);
    free(refill);
    return ret;
}

/* ------------------------------------------------------------------------------------------
 * driver
 * ---------------------------------------------------------------------------------------- */

void orc_free_result(OrcLfResult *r) {
    if (!r)
        return;
    free(r->xyb);
    free(r->dct);
    free(r->quant);
    free(r->dc);
    free(r->nz);
    free(r->symbols);
    free(r->group_symbols);
    free(r->stream);
    free(r->group_offset);
    free(r->group_bits);
    free(r);
}

OrcLfResult *orc_encode_lf_group(const void *const buf[3], ptrdiff_t row_stride, ptrdiff_t pixel_stride, int fmt,
                                 int linear_light, size_t width, size_t height, unsigned preset,
                                 unsigned num_presets, unsigned *max_alphabet_size, int *err) {
    init_tables();
    int ret = ORC_ERR_NOMEM;
    uint8_t *cmap = NULL;
    AliasTable *alias = NULL;
    OrcLfResult *r = calloc(1, sizeof(*r));
    if (!r)
        goto fail;
    r->width = width;
    r->height = height;
    r->vbw = (width + 7) >> 3;
    r->vbh = (height + 7) >> 3;
    r->stride = r->vbw << 3;
    r->gcols = (width + 255) >> 8;
    r->grows = (height + 255) >> 8;
    r->num_groups = r->gcols * r->grows;
    r->max_alphabet_size = max_alphabet_size ? *max_alphabet_size : 0;
    const size_t plane = r->stride * r->vbh * 8;
    r->xyb = malloc(3 * plane * sizeof(float));
    r->dct = malloc(3 * plane * sizeof(float));
    r->quant = malloc(3 * plane * sizeof(int32_t));
    r->dc = malloc(3 * r->vbw * r->vbh * sizeof(int32_t));
    r->nz = malloc(r->num_groups * 1024 * 3);
    r->group_symbols = calloc(r->num_groups, sizeof(size_t));
    r->group_offset = calloc(r->num_groups, sizeof(size_t));
    r->group_bits = calloc(r->num_groups, sizeof(size_t));
    cmap = malloc(1485u * (size_t)num_presets);
    alias = calloc(256, sizeof(AliasTable));
    if (!r->xyb || !r->dct || !r->quant || !r->dc || !r->nz || !r->group_symbols || !r->group_offset ||
        !r->group_bits || !cmap || !alias)
        goto fail;

    ret = stage_xyb(buf, row_stride, pixel_stride, fmt, linear_light, width, height, r->stride, r->vbh * 8, r->xyb);
    if (ret)
        goto fail;
    stage_dct(r->xyb, r->dct, r->vbw, r->vbh, r->stride);
    stage_quant(r);

    orc_hf_cluster_map(cmap, num_presets);
    r->cluster_from = cmap[1485u * preset];
    r->cluster_to = cmap[1485u * (preset + 1) - 1] + 1u;
    ret = stage_tokenize(r, preset, cmap);
    if (ret)
        goto fail;

    /* histogram of this preset's clusters (entropy.c:526-544) */
    for (size_t i = 0; i < r->num_symbols; i++)
        r->freqs[r->symbols[i].cluster][r->symbols[i].token]++;
    r->log_alphabet_size = ceil_log2_u32(r->max_alphabet_size);
    if (r->log_alphabet_size < 5)
        r->log_alphabet_size = 5;
    for (unsigned c = r->cluster_from; c < r->cluster_to; c++) {
        if (!r->alphabet_size[c])
            continue;
        const int unique = orc_normalize_frequencies(r->freqs[c], r->alphabet_size[c]);
        ret = unique < 0 ? ORC_ERR_INTERNAL : build_alias(&alias[c], r->freqs[c], r->alphabet_size[c],
                                                          r->log_alphabet_size, unique);
        if (ret)
            goto fail;
    }

    /* per-group sections */
    {
        const int preset_bits = ceil_log2_u32(num_presets);
        BitBuf all = {0};
        size_t first = 0;
        for (size_t g = 0; g < r->num_groups; g++) {
            BitBuf bb = {0};
            ret = rans_encode_group(r->symbols + first, r->group_symbols[g], r->freqs, alias, preset, preset_bits, &bb);
            if (ret) {
                free(bb.data);
                free(all.data);
                goto fail;
            }
            first += r->group_symbols[g];
            r->group_offset[g] = all.bits >> 3;
            r->group_bits[g] = bb.bits;
            for (size_t k = 0; k < (bb.bits + 7) / 8 && !ret; k++)
                ret = put_bits(&all, bb.data[k], 8);
            free(bb.data);
            if (ret) {
                free(all.data);
                goto fail;
            }
        }
        r->stream = all.data;
        r->stream_bytes = all.bits >> 3;
    }
    if (max_alphabet_size)
        *max_alphabet_size = r->max_alphabet_size;
    free(cmap);
    free(alias);
    if (err)
        *err = ORC_OK;
    return r;

fail:
    free(cmap);
    free(alias);
    orc_free_result(r);
    if (err)
        *err = ret;
    return NULL;
}

int orc_hot_path_image(const void *pixels, int fmt, size_t width, size_t height, int linear_light,
                       uint64_t *total_bytes, uint64_t *checksum) {
    const size_t sample = fmt == ORC_FMT_U8 ? 1 : fmt == ORC_FMT_U16 ? 2 : 4;
    const size_t lfx = (width + 2047) >> 11, lfy = (height + 2047) >> 11;
    const unsigned num_presets = lfx * lfy > 256 ? 256 : (unsigned)(lfx * lfy);
    unsigned max_alpha = 0;
    uint64_t bytes = 0, h = UINT64_C(0xcbf29ce484222325);
    for (size_t ty = 0; ty < lfy; ty++) {
        for (size_t tx = 0; tx < lfx; tx++) {
            const size_t x0 = tx << 11, y0 = ty << 11;
            const size_t w = x0 + 2048 > width ? width - x0 : 2048;
            const size_t hh = y0 + 2048 > height ? height - y0 : 2048;
            const uint8_t *base = (const uint8_t *)pixels + (y0 * width + x0) * 3 * sample;
            const void *buf[3] = {base, base + sample, base + 2 * sample};
            int err = 0;
            OrcLfResult *r = orc_encode_lf_group(buf, (ptrdiff_t)(3 * width), 3, fmt, linear_light, w, hh,
                                                 (unsigned)(ty * lfx + tx), num_presets, &max_alpha, &err);
            if (!r)
                return err;
            bytes += r->stream_bytes;
            for (size_t i = 0; i < r->stream_bytes; i++)
                h = (h ^ r->stream[i]) * UINT64_C(0x100000001b3);
            orc_free_result(r);
        }
    }
    if (total_bytes)
        *total_bytes = bytes;
    if (checksum)
        *checksum = h;
    return ORC_OK;
}
