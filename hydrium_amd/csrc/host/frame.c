/* frame.c — see frame.h.  Reference citations are relative to /root/reference/src/libhydrium/. */
#include "frame.h"

#include <stdlib.h>
#include <string.h>

#include "prefix.h"

#define ST_NOMEM (-13)
#define ST_API (-14)
#define ST_INTERNAL (-15)

static int ilog2_u32(uint32_t v) { return 31 - __builtin_clz(v); }
static int clog2_u64(uint64_t v) { return 63 - __builtin_clzll(v) + ((v & (v - 1)) != 0); }
static uint32_t pack_signed(int32_t v) { return ((uint32_t)v << 1) ^ (0u - ((uint32_t)v >> 31)); }

/* U32 distributions of the fields written here */
static const HydU32Dist kSizeHeader = {{1, 1, 1, 1}, {9, 13, 18, 30}};               /* encoder.c:98-101 */
static const HydU32Dist kFrameSize = {{0, 256, 2304, 18688}, {8, 11, 14, 30}};       /* encoder.c:102-105 */
static const HydU32Dist kGlobalScale = {{1, 2049, 4097, 8193}, {11, 11, 12, 16}};    /* encoder.c:106-109 */
static const HydU32Dist kQuantLf = {{16, 1, 1, 1}, {0, 5, 8, 16}};                   /* encoder.c:110-113 */
static const HydU32Dist kTocEntry = {{0, 1024, 17408, 4211712}, {10, 14, 22, 30}};   /* encoder.c:117-120 */

static const uint8_t kZeroMap[8] = {0, 0, 0, 0, 0, 0, 0, 0};

/* ---------------------------------------------------------------------------------------------
 * file header
 * ------------------------------------------------------------------------------------------- */

/* ISOBMFF prologue that announces codestream level 10 (encoder.c:23-30) */
static const uint8_t kLevel10Boxes[49] = {
    0x00, 0x00, 0x00, 0x0c, 'J', 'X', 'L', ' ', 0x0d, 0x0a, 0x87, 0x0a, 0x00, 0x00, 0x00, 0x14, 'f',
    't',  'y',  'p',  'j',  'x', 'l', ' ', 0x00, 0x00, 0x00, 0x00, 'j',  'x',  'l',  ' ',  0x00, 0x00,
    0x00, 0x09, 'j',  'x',  'l', 'l', 0x0a, 0x00, 0x00, 0x00, 0x00, 'j',  'x',  'l',  'c',
};

/* context of ICC byte i given the two bytes before it (encoder.c:122-154) */
static uint32_t icc_context(uint64_t i, uint32_t b1, uint32_t b2) {
    if (i <= 128)
        return 0;
    uint32_t p1, p2;
    const int alpha1 = (b1 >= 'a' && b1 <= 'z') || (b1 >= 'A' && b1 <= 'Z');
    const int alpha2 = (b2 >= 'a' && b2 <= 'z') || (b2 >= 'A' && b2 <= 'Z');
    const int num1 = (b1 >= '0' && b1 <= '9') || b1 == '.' || b1 == ',';
    const int num2 = (b2 >= '0' && b2 <= '9') || b2 == '.' || b2 == ',';
    if (alpha1)
        p1 = 0;
    else if (num1)
        p1 = 1;
    else if (b1 <= 1)
        p1 = b1 + 2;
    else if (b1 < 16)
        p1 = 4;
    else if (b1 > 240 && b1 < 255)
        p1 = 5;
    else if (b1 == 255)
        p1 = 6;
    else
        p1 = 7;
    if (alpha2)
        p2 = 0;
    else if (num2)
        p2 = 1;
    else if (b2 < 16)
        p2 = 2;
    else if (b2 > 240)
        p2 = 3;
    else
        p2 = 4;
    return 1 + p1 + p2 * 8;
}

int hyd_write_file_header(HydBits *out, size_t width, size_t height, int level10, const uint8_t *icc, size_t icc_size,
                          const char **err) {
    if (level10)
        hb_append_bytes(out, kLevel10Boxes, sizeof(kLevel10Boxes));
    hb_put(out, 0x0AFF, 17);          /* signature FF 0A, then div8 = 0 */
    if (hb_u32(out, &kSizeHeader, (uint32_t)height))
        return ST_API;
    hb_put(out, 0, 3);                /* ratio = 0: explicit width */
    if (hb_u32(out, &kSizeHeader, (uint32_t)width))
        return ST_API;
    hb_bool(out, 0);                  /* ImageMetadata.all_default */
    hb_bool(out, 0);                  /* extra_fields */
    hb_bool(out, 0);                  /* integer samples */
    hb_put(out, 0, 2);                /* 8 bits per sample */
    hb_bool(out, 1);                  /* modular 16-bit buffers */
    hb_put(out, 0, 2);                /* no extra channels */
    hb_bool(out, 1);                  /* xyb_encoded */
    if (icc) {
        hb_bool(out, 0);              /* ColourEncoding.all_default = 0 */
        hb_bool(out, 1);              /* want_icc */
        hb_enum(out, 0);              /* kRGB */
    } else {
        hb_bool(out, 1);
    }
    hb_u64(out, 0);                   /* extensions */
    hb_bool(out, 1);                  /* default opsin matrix */
    if (icc) {
        /* 41 contexts folded onto 9 clusters (encoder.c:156-162) */
        uint8_t map[41];
        map[0] = 0;
        for (int i = 1; i < 41; i++)
            map[i] = (uint8_t)(1 + (i - 1) % 8);
        hb_u64(out, icc_size);
        HydSymStream s;
        int ret = hps_init(&s, map, 41, 0, 0, 0);
        uint32_t b1 = 0, b2 = 0;
        for (uint64_t i = 0; i < icc_size && !ret; i++) {
            ret = hps_send(&s, icc_context(i, b1, b2), icc[i]);
            b2 = b1;
            b1 = icc[i];
        }
        if (ret) {
            hps_free(&s);
            return ret;
        }
        ret = hps_finish_prefix(&s, out, err);
        if (ret)
            return ret;
    }
    hb_align(out);
    return out->failed ? ST_NOMEM : 0;
}

/* ---------------------------------------------------------------------------------------------
 * frame header + TOC
 * ------------------------------------------------------------------------------------------- */

static size_t groups_of(size_t px) { return (px + 255) >> 8; }

size_t hyd_toc_entries(const HydFrameShape *shape) {
    const size_t groups = groups_of(shape->frame_width) * groups_of(shape->frame_height);
    return groups > 1 ? 2 + groups + shape->lfg_count : 1;
}

/* Lehmer code of the permutation that maps the spec's logical section order to our physical one
 * (encoder.c:241-325; the reference's O(n^2) scan is replaced by a Fenwick tree, same numbers) */
static int toc_lehmer(const HydFrameShape *shape, size_t n, uint32_t **lehmer_out) {
    const size_t fgx = groups_of(shape->frame_width);
    size_t *where = malloc(n * sizeof(size_t)); /* logical section -> physical position */
    const size_t unset = (size_t)-1;
    uint32_t *tree = calloc(n + 1, sizeof(uint32_t));
    uint32_t *lehmer = malloc(n * sizeof(uint32_t));
    if (!where || !tree || !lehmer) {
        free(where);
        free(tree);
        free(lehmer);
        return ST_NOMEM;
    }
    /* every logical section must be claimed exactly once: a frame description with a repeated or
     * out-of-range LF group would otherwise index past the arrays below */
    int bad = 0;
    size_t pos = 0;
    for (size_t i = 0; i < n; i++)
        where[i] = unset;
#define HYD_CLAIM(logical)                                        \
    do {                                                          \
        const size_t at_ = (logical);                             \
        if (at_ >= n || where[at_] != unset || pos >= n)          \
            bad = 1;                                              \
        else                                                      \
            where[at_] = pos++;                                   \
    } while (0)
    HYD_CLAIM(0); /* LFGlobal */
    for (size_t s = 0; s < shape->lfg_count && !bad; s++)
        HYD_CLAIM(1 + shape->lfg[s].raster_id); /* LF groups in send order */
    HYD_CLAIM(1 + shape->lfg_count);            /* HFGlobal */
    for (size_t s = 0; s < shape->lfg_count && !bad; s++) {
        const HydFrameLfg *l = &shape->lfg[s];
        const size_t gcx = groups_of(l->width), gcy = groups_of(l->height);
        const size_t gx0 = shape->one_frame ? l->x << 3 : 0, gy0 = shape->one_frame ? l->y << 3 : 0;
        for (size_t g = 0; g < gcx * gcy && !bad; g++)
            HYD_CLAIM(2 + shape->lfg_count + (gy0 + g / gcx) * fgx + gx0 + g % gcx);
    }
#undef HYD_CLAIM
    if (bad || pos != n) {
        free(where);
        free(tree);
        free(lehmer);
        return ST_API;
    }
    /* lehmer[i] = how many not-yet-used positions are smaller than where[i] */
    for (size_t i = 1; i <= n; i++) {
        tree[i] += 1;
        const size_t up = i + (i & (~i + 1));
        if (up <= n)
            tree[up] += tree[i];
    }
    for (size_t i = 0; i < n; i++) {
        uint32_t below = 0;
        for (size_t j = where[i]; j > 0; j -= j & (~j + 1))
            below += tree[j];
        lehmer[i] = below;
        for (size_t j = where[i] + 1; j <= n; j += j & (~j + 1))
            tree[j] -= 1;
    }
    free(where);
    free(tree);
    *lehmer_out = lehmer;
    return 0;
}

int hyd_write_frame_header(HydBits *out, const HydFrameShape *shape, const char **err) {
    hb_align(out);
    const int is_last = shape->one_frame || shape->is_last;
    const HydFrameLfg *tile = &shape->lfg[0];
    const int have_crop = !shape->one_frame && !(shape->image_width <= tile->width && shape->image_height <= tile->height);

    hb_put(out, 0, 1);                 /* all_default = 0 */
    hb_put(out, is_last ? 0 : 3, 2);   /* kRegularFrame, or kSkipProgressive for non-final tiles */
    hb_put(out, 0, 1);                 /* VarDCT */
    hb_u64(out, 0x80);                 /* flags: skip adaptive LF smoothing */
    hb_put(out, 0x4C, 10);             /* upsampling 0, x_qm_scale 3, b_qm_scale 2, one pass (encoder.c:352-358) */
    hb_bool(out, have_crop);
    if (have_crop) {
        const size_t tw = shape->tile_count_x << 8, th = shape->tile_count_y << 8;
        hb_u32(out, &kFrameSize, pack_signed((int32_t)(tile->x * tw)));
        hb_u32(out, &kFrameSize, pack_signed((int32_t)(tile->y * th)));
        hb_u32(out, &kFrameSize, (uint32_t)tile->width);
        hb_u32(out, &kFrameSize, (uint32_t)tile->height);
    }
    hb_put(out, 0, 2);                 /* blend mode: replace */
    if (have_crop)
        hb_put(out, 0, 2);             /* blend source 0 */
    hb_bool(out, is_last);
    if (!is_last)
        hb_put(out, 0, 2);             /* save_as_reference = 0 */
    hb_put(out, 0, 2);                 /* no name */
    hb_bool(out, 0);                   /* RestorationFilter.all_default = 0 */
    hb_bool(out, 0);                   /* no Gaborish */
    hb_put(out, 0, 2);                 /* no edge-preserving filter */
    hb_put(out, 0, 2);                 /* filter extensions */
    hb_put(out, 0, 2);                 /* frame header extensions */

    const size_t n = hyd_toc_entries(shape);
    if (n > 1) {
        hb_bool(out, 1);               /* permuted TOC */
        uint32_t *lehmer = NULL;
        int ret = toc_lehmer(shape, n, &lehmer);
        if (ret)
            return ret;
        HydSymStream s;
        ret = hps_init(&s, kZeroMap, 8, 0, 0, 0);
        if (!ret)
            ret = hps_send(&s, 0, (uint32_t)n);
        for (size_t i = 0; i < n && !ret; i++)
            ret = hps_send(&s, 0, lehmer[i]);
        free(lehmer);
        if (ret) {
            hps_free(&s);
            return ret;
        }
        ret = hps_finish_prefix(&s, out, err);
        if (ret)
            return ret;
    } else {
        hb_bool(out, 0);
    }
    hb_align(out);
    return out->failed ? ST_NOMEM : 0;
}

int hyd_write_toc_sizes(HydBits *out, const size_t *section_bytes, size_t count) {
    hb_align(out);
    for (size_t i = 0; i < count; i++)
        if (hb_u32(out, &kTocEntry, (uint32_t)section_bytes[i]))
            return ST_API; /* a section of 1 GiB or more cannot be signalled */
    hb_align(out);
    return out->failed ? ST_NOMEM : 0;
}

/* ---------------------------------------------------------------------------------------------
 * LFGlobal / LFGroup
 * ------------------------------------------------------------------------------------------- */

void hyd_write_lf_global(HydBits *out) {
    hb_bool(out, 1);                         /* LF dequant all_default */
    hb_u32(out, &kGlobalScale, 32768);       /* quantiser global scale */
    hb_u32(out, &kQuantLf, 4);               /* quant_lf */
    hb_bool(out, 0);                         /* HF block context map not default */
    hb_put(out, 0, 16);                      /* no LF / QF thresholds */
    hb_bool(out, 1);                         /* simple clustering */
    hb_put(out, 2, 2);                       /* 2 bits per entry */
    for (int c = 0; c < 3; c++)              /* 39 block contexts -> one per channel */
        for (int k = 0; k < 13; k++)
            hb_put(out, (uint64_t)c, 2);
    hb_bool(out, 1);                         /* LF channel correlation all_default */
    hb_bool(out, 0);                         /* no global modular tree */
}

/* LF group, part 1: modular header and the MA tree, a single leaf with the clamped-gradient
 * predictor (encoder.c:539-558) */
static int lf_group_prologue(HydBits *out, const char **err) {
    static const uint32_t ma_tree[5][2] = {{1, 0}, {2, 5}, {3, 0}, {4, 0}, {5, 0}}; /* encoder.c:114-116 */
    HydSymStream s;
    hb_put(out, 0, 2);   /* extra precision 0 */
    hb_bool(out, 0);     /* local tree */
    hb_bool(out, 1);     /* weighted-predictor params default */
    hb_put(out, 0, 2);   /* no transforms */
    int ret = hps_init(&s, kZeroMap, 6, 0, 0, 0);
    for (int i = 0; i < 5 && !ret; i++)
        ret = hps_send(&s, ma_tree[i][0], ma_tree[i][1]);
    if (ret) {
        hps_free(&s);
        return ret;
    }
    return hps_finish_prefix(&s, out, err);
}

/* LF group, part 3: varblock count, then a modular sub-image that says "8x8 DCT, hf_mult 5" for
 * every block and zero chroma-from-luma factors (encoder.c:598-626).  Depends on the LF group's
 * geometry only. */
static int lf_group_hf_metadata(HydBits *out, size_t vbw, size_t vbh, const char **err) {
    HydSymStream s;
    const size_t blocks = vbw * vbh;
    hb_put(out, blocks - 1, clog2_u64(blocks));
    hb_put(out, 0x2, 4);
    int ret = hps_init(&s, kZeroMap, 6, 0, 0, 0);
    for (uint32_t ctx = 1; ctx <= 5 && !ret; ctx++)
        ret = hps_send(&s, ctx, 0);
    if (ret) {
        hps_free(&s);
        return ret;
    }
    ret = hps_finish_prefix(&s, out, err);
    if (ret)
        return ret;
    const size_t cfl = ((vbw + 7) >> 3) * ((vbh + 7) >> 3);
    const size_t leading_zeros = 2 * cfl + blocks;
    ret = hps_init(&s, kZeroMap, 1, 0, 29, 1);
    for (size_t i = 0; i < leading_zeros && !ret; i++)
        ret = hps_send(&s, 0, 0);
    for (size_t i = 0; i < blocks && !ret; i++)
        ret = hps_send(&s, 0, (5 - 1) * 2); /* pack_signed(hf_mult - 1) */
    for (size_t i = 0; i < blocks && !ret; i++)
        ret = hps_send(&s, 0, 0);
    if (ret) {
        hps_free(&s);
        return ret;
    }
    return hps_finish_prefix(&s, out, err);
}

int hyd_write_lf_group(HydBits *out, const int32_t *dc, size_t vbw, size_t vbh, const char **err) {
    HydSymStream s;
    int ret = lf_group_prologue(out, err);
    if (ret)
        return ret;

    /* LF coefficients, channel order Y, X, B (encoder.c:574-594); the ints themselves come from
     * the transform kernel */
    const size_t blocks = vbw * vbh;
    ret = hps_init(&s, kZeroMap, 1, 1, HYD_LF_RUN_BASE, 1);
    if (ret)
        return ret;
    hps_set_config(&s, 0, 0, 7, 1, 1);
    for (int i = 0; i < 3 && !ret; i++) {
        const int32_t *p = dc + (size_t)(i < 2 ? 1 - i : i) * blocks;
        for (size_t y = 0; y < vbh && !ret; y++) {
            const int32_t *row = p + y * vbw, *above = row - vbw;
            for (size_t x = 0; x < vbw && !ret; x++) {
                const int32_t w = x ? row[x - 1] : y ? above[x] : 0;
                const int32_t n = y ? above[x] : w;
                const int32_t nw = x && y ? above[x - 1] : w;
                const int32_t lo = w < n ? w : n, hi = w < n ? n : w;
                int32_t pred = w + n - nw;
                pred = pred < lo ? lo : pred > hi ? hi : pred;
                ret = hps_send(&s, 0, pack_signed(row[x] - pred));
            }
        }
    }
    if (ret) {
        hps_free(&s);
        return ret;
    }
    ret = hps_finish_prefix(&s, out, err);
    if (ret)
        return ret;
    ret = lf_group_hf_metadata(out, vbw, vbh, err);
    if (ret)
        return ret;
    return out->failed ? ST_NOMEM : 0;
}

int hyd_write_lf_group_tail(HydBits *out, size_t vbw, size_t vbh, const char **err) {
    return lf_group_hf_metadata(out, vbw, vbh, err);
}

int hyd_write_lf_group_coded(HydBits *out, size_t vbw, size_t vbh, const HydLfCoded *lf, const HydBits *tail,
                             const char **err) {
    int ret = lf_group_prologue(out, err);
    if (ret)
        return ret;
    /* the LF-coefficient stream: the same layout hyd_write_lf_group sets up (one value context plus
     * the LZ77 distance context, both with config (7,1,1)); lengths and symbol bits come from the GPU */
    if (lf->alphabet < 1 || lf->alphabet > HYD_LF_RUN_BASE + 128u || lf->bit_count > ((uint64_t)3 * vbw * vbh) * 64) {
        if (err)
            *err = "LF stream from the device is malformed";
        return ST_INTERNAL;
    }
    static const uint8_t map[2] = {0, 1};
    static const HydUintConfig config[2] = {{7, 1, 1}, {7, 1, 1}};
    const uint16_t alphabet[2] = {(uint16_t)lf->alphabet, (uint16_t)(lf->run_pairs ? 2 : 0)};
    /* (hyd_write_lf_group_fixed_head below writes the same layout's fixed fields for the device-side assembler) */
    uint32_t *lengths = calloc((size_t)alphabet[0] + alphabet[1], sizeof(uint32_t));
    if (!lengths)
        return ST_NOMEM;
    for (uint32_t t = 0; t < lf->alphabet; t++) {
        if (t < 256)
            lengths[t] = lf->lengths[t];
        else if (t >= HYD_LF_RUN_BASE)
            lengths[t] = lf->lengths[256 + (t - HYD_LF_RUN_BASE)];
    }
    const HydPrefixLayout lay = {HYD_LF_RUN_BASE, 3, map, 2, 2, config, alphabet};
    ret = hps_write_header(out, &lay, lengths, err);
    free(lengths);
    if (ret)
        return ret;
    hb_append_bits(out, lf->bits, lf->bit_count);
    if (tail) { /* the geometry-only HF metadata, coded once per distinct LF-group shape by the caller */
        hb_append_bits(out, tail->data, (uint64_t)tail->len * 8);
        if (tail->nacc)
            hb_put(out, tail->acc, tail->nacc);
    } else {
        ret = lf_group_hf_metadata(out, vbw, vbh, err);
        if (ret)
            return ret;
    }
    return out->failed ? ST_NOMEM : 0;
}

/* Everything of an LF group section in front of its first data-dependent bit: modular header, MA tree
 * and the fixed fields of the LF-coefficient stream's header (the layout hyd_write_lf_group_coded uses).
 * The device-side assembler (csrc/hip/assemble.hip) continues from there with hydk_lf_prefix_codes. */
int hyd_write_lf_group_fixed_head(HydBits *out, const char **err) {
    int ret = lf_group_prologue(out, err);
    if (ret)
        return ret;
    static const uint8_t map[2] = {0, 1};
    static const HydUintConfig config[2] = {{7, 1, 1}, {7, 1, 1}};
    static const uint16_t alphabet[2] = {0, 0};
    const HydPrefixLayout lay = {HYD_LF_RUN_BASE, 3, map, 2, 2, config, alphabet};
    ret = hps_write_header_fixed(out, &lay, err);
    if (ret)
        return ret;
    return out->failed ? ST_NOMEM : 0;
}

/* ---------------------------------------------------------------------------------------------
 * HFGlobal
 * ------------------------------------------------------------------------------------------- */

int hyd_hf_cluster_map(uint8_t *map, unsigned num_presets) {
    const int per = num_presets * 9 <= 256 ? 9 : num_presets * 3 <= 256 ? 3 : num_presets * 2 <= 256 ? 2 : 1;
    for (unsigned p = 0; p < num_presets; p++) {
        uint8_t *m = map + 1485u * p;
        for (unsigned ctx = 0; ctx < 1485; ctx++) {
            unsigned local;
            if (ctx < 111) /* non-zero-count contexts */
                local = per == 9 ? ctx % 3 : 0;
            else           /* coefficient contexts */
                local = per == 9 ? 3 + (ctx - 111) % 6 : per == 3 ? 1 + ((ctx - 111) & 1) : per == 2 ? 1 : 0;
            m[ctx] = (uint8_t)(per * p + local);
        }
    }
    return per;
}

static void put_ans_u8(HydBits *out, uint32_t v) { /* entropy.c:71-78 */
    hb_bool(out, v != 0);
    if (!v)
        return;
    const int l = ilog2_u32(v);
    hb_put(out, (uint64_t)l, 3);
    hb_put(out, v, l);
}

/* one 12-bit distribution in the ANS histogram syntax (entropy.c:303-369) */
static void put_ans_distribution(HydBits *out, const uint32_t *freq, uint32_t alphabet) {
    static const uint8_t log_code[14][2] = {{17, 5}, {11, 4}, {15, 4}, {3, 4}, {9, 4}, {7, 4}, {4, 3},
                                            {2, 3},  {5, 3},  {6, 3},  {0, 3}, {33, 6}, {1, 7}, {65, 7}}; /* entropy.c:35-38 */
    if (!alphabet) {
        hb_put(out, 1, 2); /* an unused cluster is sent as "always symbol 0" */
        put_ans_u8(out, 0);
        return;
    }
    int first = -1, second = -1, seen = 0;
    for (uint32_t k = 0; k < alphabet; k++) {
        if (freq[k] == 4096) {
            hb_put(out, 1, 2);
            put_ans_u8(out, k);
            return;
        }
        if (!freq[k])
            continue;
        if (++seen > 2)
            break;
        if (first < 0) {
            first = (int)k;
        } else if (freq[first] + freq[k] == 4096) {
            second = (int)k;
            break;
        }
    }
    if (first >= 0 && second >= 0) {
        hb_put(out, 3, 2); /* two symbols */
        put_ans_u8(out, (uint32_t)first);
        put_ans_u8(out, (uint32_t)second);
        hb_put(out, freq[first], 12);
        return;
    }
    hb_put(out, 0, 2);  /* neither simple nor flat */
    hb_put(out, 0x7, 3);
    hb_put(out, 0x6, 3); /* together: shift = 13 */
    put_ans_u8(out, alphabet - 3);
    int logc[HYD_FRAME_ALPHABET];
    uint32_t omit = 0;
    int omit_log = 0;
    for (uint32_t k = 0; k < alphabet; k++) {
        logc[k] = freq[k] ? 1 + ilog2_u32(freq[k]) : 0;
        hb_put(out, log_code[logc[k]][0], log_code[logc[k]][1]);
        if (logc[k] > omit_log) {
            omit_log = logc[k];
            omit = k;
        }
    }
    for (uint32_t k = 0; k < alphabet; k++) {
        if (k == omit || logc[k] <= 1)
            continue;
        hb_put(out, freq[k], logc[k] - 1);
    }
}

/* HFGlobal up to and including "ANS, not prefix codes": depends on the frame's geometry only */
int hyd_write_hf_global_fixed(HydBits *out, unsigned num_presets, size_t num_frame_groups, int *clusters_per_preset,
                              const char **err) {
    hb_bool(out, 1);                                                 /* default dequant matrices */
    hb_put(out, num_presets - 1, clog2_u64(num_frame_groups));       /* number of HF presets */
    hb_put(out, 2, 2);                                               /* coefficient order: default */

    /* ANS stream header (entropy.c:980-1001) */
    uint8_t *map = malloc((size_t)1485 * num_presets);
    if (!map)
        return ST_NOMEM;
    const int per = hyd_hf_cluster_map(map, num_presets);
    const size_t num_clusters = (size_t)per * num_presets;
    hb_bool(out, 0); /* no LZ77 */
    int ret = hps_write_cluster_map(map, (size_t)1485 * num_presets, num_clusters, out, err);
    free(map);
    if (ret)
        return ret;
    hb_bool(out, 0); /* ANS, not prefix codes */
    if (clusters_per_preset)
        *clusters_per_preset = per;
    return out->failed ? ST_NOMEM : 0;
}

int hyd_write_hf_global(HydBits *out, unsigned num_presets, size_t num_frame_groups,
                        const uint32_t (*freq)[HYD_FRAME_MAX_CLUSTERS][HYD_FRAME_ALPHABET],
                        const uint32_t (*alphabet)[HYD_FRAME_MAX_CLUSTERS], unsigned max_alphabet, const char **err) {
    int per = 0;
    int ret = hyd_write_hf_global_fixed(out, num_presets, num_frame_groups, &per, err);
    if (ret)
        return ret;
    const size_t num_clusters = (size_t)per * num_presets;
    int log_alpha = max_alphabet > 1 ? clog2_u64(max_alphabet) : 0;
    if (log_alpha < 5)
        log_alpha = 5;
    hb_put(out, (uint64_t)(log_alpha - 5), 2);
    const HydUintConfig cfg = {4, 1, 0}; /* encoder.c:908 */
    for (size_t c = 0; c < num_clusters; c++)
        hps_write_uint_config(out, &cfg, log_alpha);
    for (unsigned p = 0; p < num_presets; p++)
        for (int c = 0; c < per; c++)
            put_ans_distribution(out, freq[p][c], alphabet[p][c]);
    return out->failed ? ST_NOMEM : 0;
}

/* ---------------------------------------------------------------------------------------------
 * CPU-only test hooks (libhydrium_hosttest.so): the field writers the device-side assembler runs
 * (csrc/hip/hydk_sections.h, compiled here for the host) next to the host functions they restate.
 * ------------------------------------------------------------------------------------------- */
#ifdef HYD_TEST_HOOKS
#include "../hip/hydk_sections.h"
#define HYDT_EXPORT __attribute__((visibility("default")))

static int hydt_take_bits(HydBits *b, uint8_t *out, size_t cap, uint64_t *nbits) {
    *nbits = hb_bit_count(b);
    hb_align(b);
    const int ok = !b->failed && b->len <= cap;
    if (ok)
        memcpy(out, b->data, b->len);
    hb_free(b);
    return ok ? 0 : ST_NOMEM;
}

/* an LF group section up to its first symbol bit, by the host path (hyd_write_lf_group_coded's first half) ... */
HYDT_EXPORT int hydt_lf_head_host(const uint8_t *lengths, uint32_t alphabet0, uint32_t run_pairs, uint8_t *out, size_t cap,
                                  uint64_t *nbits) {
    HydBits b;
    const char *err = NULL;
    hb_init(&b);
    int ret = lf_group_prologue(&b, &err);
    if (ret)
        return ret;
    static const uint8_t map[2] = {0, 1};
    static const HydUintConfig config[2] = {{7, 1, 1}, {7, 1, 1}};
    const uint16_t alphabet[2] = {(uint16_t)alphabet0, (uint16_t)(run_pairs ? 2 : 0)};
    uint32_t *len = calloc((size_t)alphabet[0] + alphabet[1] + 1, sizeof(uint32_t));
    if (!len)
        return ST_NOMEM;
    for (uint32_t t = 0; t < alphabet0; t++) {
        if (t < 256)
            len[t] = lengths[t];
        else if (t >= HYD_LF_RUN_BASE)
            len[t] = lengths[256 + (t - HYD_LF_RUN_BASE)];
    }
    const HydPrefixLayout lay = {HYD_LF_RUN_BASE, 3, map, 2, 2, config, alphabet};
    ret = hps_write_header(&b, &lay, len, &err);
    free(len);
    if (ret) {
        hb_free(&b);
        return ret;
    }
    return hydt_take_bits(&b, out, cap, nbits);
}

/* ... and by the assembler's: fixed head from the plan + hydk_lf_prefix_codes */
HYDT_EXPORT int hydt_lf_head_sections(const uint8_t *lengths, uint32_t alphabet0, uint32_t run_pairs, uint8_t *out, size_t cap,
                                      uint64_t *nbits) {
    HydBits b;
    const char *err = NULL;
    hb_init(&b);
    int ret = hyd_write_lf_group_fixed_head(&b, &err);
    if (ret)
        return ret;
    const uint64_t fixed = hb_bit_count(&b);
    hb_align(&b);
    uint32_t *words = calloc(cap / 4 + 2, sizeof(uint32_t));
    if (!words) {
        hb_free(&b);
        return ST_NOMEM;
    }
    memcpy(words, b.data, b.len);
    hb_free(&b);
    HydkSink sink = {words, fixed, (uint64_t)(cap / 4) * 32u, 0, 0};
    ret = hydk_lf_prefix_codes(&sink, lengths, alphabet0, run_pairs);
    if (!ret && sink.overflow)
        ret = ST_NOMEM;
    *nbits = sink.pos;
    if (!ret)
        memcpy(out, words, (size_t)((sink.pos + 7) >> 3));
    free(words);
    return ret;
}

/* ... and by the wavefront form of the same writer (on the host its 64 lanes run one after another) */
HYDT_EXPORT int hydt_lf_head_wave(const uint8_t *lengths, uint32_t alphabet0, uint32_t run_pairs, uint8_t *out, size_t cap,
                                  uint64_t *nbits) {
    HydBits b;
    const char *err = NULL;
    hb_init(&b);
    int ret = hyd_write_lf_group_fixed_head(&b, &err);
    if (ret)
        return ret;
    const uint64_t fixed = hb_bit_count(&b);
    hb_align(&b);
    uint32_t *words = calloc(cap / 4 + 2, sizeof(uint32_t));
    HydkLfHeadScratch *scratch = calloc(1, sizeof(*scratch));
    if (!words || !scratch) {
        free(words);
        free(scratch);
        hb_free(&b);
        return ST_NOMEM;
    }
    memcpy(words, b.data, b.len);
    hb_free(&b);
    uint64_t end = 0;
    ret = hydk_lf_prefix_codes_wave(words, (uint64_t)(cap / 4) * 32u, fixed, lengths, alphabet0, run_pairs, scratch, &end);
    *nbits = end;
    if (!ret)
        memcpy(out, words, (size_t)((end + 7) >> 3));
    free(words);
    free(scratch);
    return ret;
}

HYDT_EXPORT int hydt_ans_distribution_host(const uint32_t *freq, uint32_t alphabet, uint8_t *out, size_t cap, uint64_t *nbits) {
    HydBits b;
    hb_init(&b);
    put_ans_distribution(&b, freq, alphabet);
    return hydt_take_bits(&b, out, cap, nbits);
}

HYDT_EXPORT int hydt_ans_distribution_sections(const uint32_t *freq, uint32_t alphabet, uint8_t *out, size_t cap, uint64_t *nbits) {
    uint32_t *words = calloc(cap / 4 + 2, sizeof(uint32_t));
    if (!words)
        return ST_NOMEM;
    HydkSink sink = {words, 0, (uint64_t)(cap / 4) * 32u, 0, 0};
    hydk_put_ans_distribution(&sink, freq, alphabet);
    HydkSink count = {NULL, 0, ~UINT64_C(0), 0, 0};
    hydk_put_ans_distribution(&count, freq, alphabet);
    const int ret = sink.overflow || count.pos != sink.pos ? ST_INTERNAL : 0;
    *nbits = sink.pos;
    if (!ret)
        memcpy(out, words, (size_t)((sink.pos + 7) >> 3));
    free(words);
    return ret;
}

/* one TOC entry both ways: returns 0 when they agree */
HYDT_EXPORT int hydt_toc_entry_check(uint64_t size) {
    HydBits b;
    hb_init(&b);
    const int host_fail = size > UINT32_MAX || hb_u32(&b, &kTocEntry, (uint32_t)size);
    const uint64_t nbits = hb_bit_count(&b);
    hb_align(&b);
    uint64_t v = 0, hv = 0;
    const uint32_t w = hydk_toc_entry(size, &v);
    for (size_t i = 0; i < b.len && i < 8; i++)
        hv |= (uint64_t)b.data[i] << (8 * i);
    hb_free(&b);
    if (host_fail)
        return w == 0 ? 0 : 1;
    return w == nbits && v == hv ? 0 : 1;
}

HYDT_EXPORT int hydt_small_code_lengths(const uint32_t *freq, uint32_t *lengths, uint32_t n, int max_depth) {
    return n <= HYDK_SMALL_N ? hydk_small_code_lengths(freq, lengths, n, max_depth) : ST_API;
}
#endif /* HYD_TEST_HOOKS */
