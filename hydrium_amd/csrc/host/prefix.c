/* prefix.c — see prefix.h.  Reference citations are relative to /root/reference/src/libhydrium/. */
#include "prefix.h"

#include <stdlib.h>
#include <string.h>

#define ST_NOMEM (-13)
#define ST_INTERNAL (-15)

static int ilog2_u32(uint32_t v) { return 31 - __builtin_clz(v); }
static int clog2_u32(uint32_t v) { return ilog2_u32(v) + ((v & (v - 1)) != 0); }

static const HydUintConfig kRunLengthConfig = {7, 0, 0};                      /* entropy.c:40 */
static const HydU32Dist kRleMinSymbol = {{224, 512, 4096, 8}, {0, 0, 0, 15}}; /* entropy.c:48-51 */
static const HydU32Dist kRleMinLength = {{3, 4, 5, 9}, {0, 0, 2, 8}};         /* entropy.c:52-55 */

/* ---------------------------------------------------------------------------------------------
 * symbol buffering
 * ------------------------------------------------------------------------------------------- */

void hps_hybridize(uint32_t value, const HydUintConfig *cfg, HydSym *out) { /* entropy.c:427-444 */
    const uint32_t split = 1u << cfg->split_exponent;
    if (value < split) {
        out->token = (uint16_t)value;
        out->residue = 0;
        out->residue_bits = 0;
        return;
    }
    const int lsb = cfg->lsb_in_token, msb = cfg->msb_in_token;
    const int n = ilog2_u32(value) - lsb - msb;
    const uint32_t low = value & ((1u << lsb) - 1u);
    uint32_t rest = value >> lsb;
    out->residue = rest & (n >= 32 ? ~0u : (1u << n) - 1u);
    rest >>= n;
    const uint32_t high = rest & ((1u << msb) - 1u);
    out->residue_bits = (uint8_t)n;
    out->token = (uint16_t)(split + (low | (high << lsb) | ((uint32_t)(n - cfg->split_exponent + lsb + msb) << (msb + lsb))));
}

void hps_set_config(HydSymStream *s, uint8_t from_cluster, uint8_t to_cluster, int split_exponent, int msb, int lsb) {
    /* entropy.c:91-106; to_cluster == 0 means "through the last cluster" */
    for (size_t j = from_cluster; (!to_cluster || j < to_cluster) && j < s->num_clusters; j++) {
        s->config[j].split_exponent = (uint8_t)split_exponent;
        s->config[j].msb_in_token = (uint8_t)msb;
        s->config[j].lsb_in_token = (uint8_t)lsb;
    }
}

void hps_free(HydSymStream *s) {
    free(s->cluster_map);
    free(s->sym);
    memset(s, 0, sizeof(*s));
}

int hps_init(HydSymStream *s, const uint8_t *cluster_map, size_t num_dists, int custom_configs,
             uint32_t rle_min_symbol, int modular) { /* entropy.c:371-425 */
    memset(s, 0, sizeof(*s));
    if (!num_dists)
        return ST_INTERNAL;
    const size_t plain = num_dists;
    if (rle_min_symbol) {
        num_dists++;
        s->rle_min_length = 3;
        s->rle_min_symbol = rle_min_symbol;
    }
    s->num_dists = num_dists;
    s->modular = modular;
    s->cluster_map = malloc(num_dists);
    if (!s->cluster_map)
        return ST_NOMEM;
    memcpy(s->cluster_map, cluster_map, plain);
    for (size_t i = 0; i < plain; i++)
        if (s->cluster_map[i] >= s->num_clusters)
            s->num_clusters = (size_t)s->cluster_map[i] + 1;
    if (s->num_clusters > num_dists) {
        hps_free(s);
        return ST_INTERNAL;
    }
    if (rle_min_symbol)
        s->cluster_map[num_dists - 1] = (uint8_t)s->num_clusters++;
    if (!custom_configs) {
        hps_set_config(s, 0, (uint8_t)(s->num_clusters - (rle_min_symbol ? 1 : 0)), 4, 1, 1);
        if (rle_min_symbol)
            hps_set_config(s, (uint8_t)(s->num_clusters - 1), (uint8_t)s->num_clusters, 7, 0, 0);
    }
    return 0;
}

static int push_symbol(HydSymStream *s, const HydSym *sym) { /* entropy.c:446-464 */
    if (s->count == s->cap) {
        const size_t ncap = s->cap ? s->cap * 2 : 1024;
        HydSym *ns = realloc(s->sym, ncap * sizeof(HydSym));
        if (!ns) {
            s->failed = 1;
            return ST_NOMEM;
        }
        s->sym = ns;
        s->cap = ncap;
    }
    s->sym[s->count++] = *sym;
    if (sym->token + 1u > s->max_alphabet)
        s->max_alphabet = (uint16_t)(sym->token + 1u);
    if (sym->token + 1u > s->alphabet[sym->cluster])
        s->alphabet[sym->cluster] = (uint16_t)(sym->token + 1u);
    return 0;
}

static int send_literal(HydSymStream *s, size_t dist, uint32_t value) { /* entropy.c:466-471 */
    HydSym sym;
    sym.cluster = s->cluster_map[dist];
    hps_hybridize(value, &s->config[sym.cluster], &sym);
    return push_symbol(s, &sym);
}

static int flush_run(HydSymStream *s) { /* entropy.c:473-500 */
    int ret = 0;
    if (s->run > s->rle_min_length) {
        HydSym sym;
        hps_hybridize(s->run - s->rle_min_length, &kRunLengthConfig, &sym);
        sym.cluster = s->cluster_map[s->last_dist];
        sym.token = (uint16_t)(sym.token + s->rle_min_symbol);
        ret = push_symbol(s, &sym);
        if (!ret) /* distance symbol: "the previous sample" in either numbering */
            ret = send_literal(s, s->num_dists - 1, s->modular ? 1u : 0u);
    } else if (s->last_value_plus1 && s->run) {
        for (uint32_t k = 0; k < s->run && !ret; k++)
            ret = send_literal(s, s->last_dist, s->last_value_plus1 - 1);
    }
    s->run = 0;
    return ret;
}

int hps_send(HydSymStream *s, size_t dist, uint32_t value) { /* entropy.c:502-524 */
    if (!s->rle_min_symbol)
        return send_literal(s, dist, value);
    if (s->last_value_plus1 == value + 1 && s->cluster_map[s->last_dist] == s->cluster_map[dist] && s->run < 127) {
        s->run++;
        return 0;
    }
    int ret = flush_run(s);
    if (ret)
        return ret;
    s->last_value_plus1 = value + 1;
    s->last_dist = (uint32_t)dist;
    return send_literal(s, dist, value);
}

/* ---------------------------------------------------------------------------------------------
 * shared stream-header fields
 * ------------------------------------------------------------------------------------------- */

void hps_write_uint_config(HydBits *out, const HydUintConfig *cfg, int log_alphabet_size) { /* entropy.c:169-182 */
    hb_put(out, cfg->split_exponent, clog2_u32(1u + (uint32_t)log_alphabet_size));
    if (cfg->split_exponent == log_alphabet_size)
        return;
    hb_put(out, cfg->msb_in_token, clog2_u32(1u + cfg->split_exponent));
    hb_put(out, cfg->lsb_in_token, clog2_u32(1u + cfg->split_exponent - cfg->msb_in_token));
}

int hps_write_cluster_map(const uint8_t *map, size_t num_dists, size_t num_clusters, HydBits *out, const char **err) {
    /* entropy.c:108-167 */
    if (num_dists == 1)
        return 0;
    const int nbits = clog2_u32((uint32_t)num_clusters);
    if (nbits <= 3 && num_dists * (size_t)nbits <= 32) {
        hb_bool(out, 1);
        hb_put(out, (uint64_t)nbits, 2);
        for (size_t i = 0; i < num_dists; i++)
            hb_put(out, map[i], nbits);
        return 0;
    }
    hb_bool(out, 0); /* not the simple form */
    hb_bool(out, 1); /* move-to-front */
    static const uint8_t one_cluster[1] = {0};
    HydSymStream nested;
    int ret = hps_init(&nested, one_cluster, 1, 1, 64, 0);
    if (ret)
        return ret;
    hps_set_config(&nested, 0, 0, 4, 1, 0);
    uint8_t mtf[256];
    for (int i = 0; i < 256; i++)
        mtf[i] = (uint8_t)i;
    for (size_t j = 0; j < num_dists && !ret; j++) {
        int index = 0;
        while (mtf[index] != map[j])
            index++;
        ret = hps_send(&nested, 0, (uint32_t)index);
        if (index) {
            const uint8_t v = mtf[index];
            memmove(mtf + 1, mtf, (size_t)index);
            mtf[0] = v;
        }
    }
    if (ret) {
        hps_free(&nested);
        return ret;
    }
    return hps_finish_prefix(&nested, out, err);
}

/* ---------------------------------------------------------------------------------------------
 * depth-limited Huffman code lengths with the reference's selection order (entropy.c:577-662)
 * ------------------------------------------------------------------------------------------- */

typedef struct HuffNode {
    uint32_t freq;
    int32_t token;     /* symbol + 1 for leaves, 0 for merged nodes */
    int32_t depth;     /* number of merges above this node (+1 for a merged node itself) */
    int32_t deepest;   /* largest depth value inside this node's subtree */
    int32_t left, right;
} HuffNode;

/* a orders before b: smaller frequency first; at equal frequency leaves before merged nodes, leaves by
 * symbol, and a merged node loses to whatever it is compared against (entropy.c:577-581) */
static int node_before(const HuffNode *a, const HuffNode *b) {
    int32_t d;
    if (a->freq != b->freq)
        d = !b->freq ? -1 : !a->freq ? 1 : (int32_t)(a->freq - b->freq);
    else
        d = !b->token ? -1 : !a->token ? 1 : a->token - b->token;
    return d < 0;
}

static int32_t deepen(HuffNode *nodes, int32_t i) { /* entropy.c:583-590 */
    if (i < 0)
        return 0;
    const int32_t self = ++nodes[i].depth;
    const int32_t l = deepen(nodes, nodes[i].left);
    const int32_t r = deepen(nodes, nodes[i].right);
    int32_t m = self > l ? self : l;
    if (r > m)
        m = r;
    return nodes[i].deepest = m;
}

int hps_code_lengths(const uint32_t *freq, uint32_t *lengths, uint32_t n, int max_depth) {
    HuffNode *nodes = calloc(2 * (size_t)n - 1, sizeof(HuffNode));
    if (!nodes)
        return ST_NOMEM;
    uint32_t live_count = 0;
    for (uint32_t i = 0; i < n; i++) {
        nodes[i].freq = freq[i];
        nodes[i].token = (int32_t)i + 1;
        nodes[i].left = nodes[i].right = -1;
        live_count += freq[i] != 0;
    }
    int ret = 0;
    if (!live_count) {
        ret = ST_INTERNAL;
        goto done;
    }
    if (max_depth < 0)
        max_depth = clog2_u32(n + 1);
    /* Round k settles slots 2k and 2k+1 as the children of merged node n+k.  Candidates are the
     * unsettled slots [2k, n+k) with non-zero weight whose subtree may still grow one level
     * without the final tree exceeding max_depth.  The reference walks every slot each round;
     * with LZ77 length tokens starting at 16384 almost all of them have weight zero, so we keep
     * the weighted slots in an ascending list instead — same visiting order, same result. */
    uint32_t *live = malloc((2 * (size_t)n) * sizeof(uint32_t));
    if (!live) {
        ret = ST_NOMEM;
        goto done;
    }
    size_t nlive = 0;
    for (uint32_t i = 0; i < n; i++)
        if (nodes[i].freq)
            live[nlive++] = i;
    for (uint32_t k = 0; k + 1 < n; k++, live_count--) {
        const int32_t limit = max_depth - clog2_u32(live_count) + 1;
        int32_t first = -1, second = -1;
        for (size_t li = 0; li < nlive; li++) {
            const uint32_t j = live[li];
            if (nodes[j].deepest >= limit)
                continue;
            if (first < 0 || node_before(&nodes[j], &nodes[first])) {
                second = first;
                first = (int32_t)j;
            } else if (second < 0 || node_before(&nodes[j], &nodes[second])) {
                second = (int32_t)j;
            }
        }
        if (first < 0) {
            ret = ST_INTERNAL;
            break;
        }
        HuffNode tmp = nodes[first];
        nodes[first] = nodes[2 * k];
        nodes[2 * k] = tmp;
        if (second < 0)
            break; /* a single tree is left */
        if ((uint32_t)second == 2 * k)
            second = first; /* it was just moved out of slot 2k */
        tmp = nodes[second];
        nodes[second] = nodes[2 * k + 1];
        nodes[2 * k + 1] = tmp;
        HuffNode *parent = &nodes[n + k];
        parent->freq = nodes[2 * k].freq + nodes[2 * k + 1].freq;
        parent->left = (int32_t)(2 * k);
        parent->right = (int32_t)(2 * k + 1);
        deepen(nodes, (int32_t)(n + k));
        /* rebuild the ascending list: slots 2k and 2k+1 are settled; the slots the two picks came
         * from now hold whatever used to sit in 2k / 2k+1 (possibly weightless); n+k is new */
        size_t w = 0;
        for (size_t li = 0; li < nlive; li++) {
            const uint32_t j = live[li];
            if (j > 2 * k + 1 && nodes[j].freq)
                live[w++] = j;
        }
        /* a pick that came from beyond the list's old entries cannot exist (picks are list members);
         * but a displaced weighted node may have landed in a slot that was not in the list before
         * only if that slot was a pick, which the filter above has kept when it now has weight */
        live[w++] = n + k;
        nlive = w;
    }
    free(live);
    if (ret)
        goto done;
    for (uint32_t j = 0; j < 2 * n - 1; j++)
        if (nodes[j].token)
            lengths[nodes[j].token - 1] = (uint32_t)nodes[j].depth;
done:
    free(nodes);
    return ret;
}

/* ---------------------------------------------------------------------------------------------
 * canonical codes and their transmission
 * ------------------------------------------------------------------------------------------- */

typedef struct Code {
    uint32_t bits; /* already bit-reversed for an LSB-first writer */
    uint32_t len;
} Code;

static uint32_t reverse_bits(uint32_t v, uint32_t len) {
    uint32_t r = 0;
    for (uint32_t i = 0; i < len; i++)
        r |= ((v >> i) & 1u) << (len - 1 - i);
    return r;
}

/* canonical assignment: shorter codes first, ties by symbol (entropy.c:664-707) */
static int assign_codes(Code *table, const uint32_t *lengths, uint32_t n) {
    uint64_t next = 0; /* code value left-aligned in 32 bits */
    for (uint32_t len = 1; len <= 32; len++) {
        for (uint32_t sidx = 0; sidx < n; sidx++) {
            if (lengths[sidx] != len)
                continue;
            table[sidx].bits = reverse_bits((uint32_t)(next >> (32 - len)), len);
            table[sidx].len = len;
            next += UINT64_C(1) << (32 - len);
        }
    }
    if (next && next != (UINT64_C(1) << 32))
        return ST_INTERNAL; /* lengths do not form a complete code */
    return 0;
}

static const uint32_t kLengthOrder[18] = {1, 2, 3, 4, 0, 5, 17, 6, 16, 7, 8, 9, 10, 11, 12, 13, 14, 15}; /* entropy.c:42 */
static const Code kLengthOfLength[6] = {{0, 2}, {7, 4}, {3, 3}, {2, 2}, {1, 2}, {15, 4}};                /* entropy.c:44-46 */

static void put_zero_run(HydBits *out, const Code *l1, uint32_t run) { /* entropy.c:709-728 */
    if (run >= 3) {
        uint32_t digits[8];
        int nd = 0;
        while (run > 10) {
            const uint32_t shorter = (run + 13) / 8;
            digits[nd++] = run - 8 * shorter + 16;
            run = shorter;
        }
        digits[nd++] = run;
        while (nd--) {
            hb_put(out, l1[17].bits, (int)l1[17].len);
            hb_put(out, digits[nd] - 3, 3);
        }
    } else {
        for (uint32_t k = 0; k < run; k++)
            hb_put(out, l1[0].bits, (int)l1[0].len);
    }
}

static int put_complex_lengths(HydBits *out, uint32_t n, const uint32_t *lengths) { /* entropy.c:730-805 */
    hb_put(out, 0, 2); /* hskip = 0 */
    uint32_t l1_freq[18] = {0};
    uint32_t run = 0;
    for (uint32_t j = 0; j < n; j++) {
        if (!lengths[j]) {
            run++;
            continue;
        }
        if (run >= 3) {
            while (run > 10) {
                l1_freq[17]++;
                run = (run + 13) / 8;
            }
            l1_freq[17]++;
        } else {
            l1_freq[0] += run;
        }
        run = 0;
        l1_freq[lengths[j]]++;
    }
    uint32_t l1_len[18] = {0};
    int ret = hps_code_lengths(l1_freq, l1_len, 18, 5);
    if (ret)
        return ret;
    uint32_t space = 0;
    for (int j = 0; j < 18; j++) {
        const uint32_t len = l1_len[kLengthOrder[j]];
        hb_put(out, kLengthOfLength[len].bits, (int)kLengthOfLength[len].len);
        if (len)
            space += 32u >> len;
        if (space >= 32)
            break;
    }
    if (space && space != 32)
        return ST_INTERNAL;
    Code l1[18];
    memset(l1, 0, sizeof(l1));
    ret = assign_codes(l1, l1_len, 18);
    if (ret)
        return ret;
    space = 0;
    run = 0;
    for (uint32_t j = 0; j < n; j++) {
        const uint32_t len = lengths[j];
        if (!len) {
            run++;
            continue;
        }
        put_zero_run(out, l1, run);
        run = 0;
        hb_put(out, l1[len].bits, (int)l1[len].len);
        space += 32768u >> len;
        if (space == 32768)
            break;
    }
    put_zero_run(out, l1, run);
    return 0;
}

typedef struct Pick {
    uint32_t symbol, len;
} Pick;

static void swap_pick(Pick *a, Pick *b) {
    const Pick t = *a;
    *a = *b;
    *b = t;
}

/* Everything of a prefix-coded stream that precedes its symbols: the fields shared with the ANS
 * header (entropy.c:546-575, log_alphabet_size = 0), the alphabet sizes (entropy.c:835-844) and one
 * code per cluster in the simple or the complex form (entropy.c:846-927).  `lengths` holds the code
 * lengths of cluster c at offset sum(alphabet[0..c)). */
int hps_write_header_fixed(HydBits *out, const HydPrefixLayout *lay, const char **err) {
    hb_bool(out, lay->rle_min_symbol != 0);
    if (lay->rle_min_symbol) {
        hb_u32(out, &kRleMinSymbol, lay->rle_min_symbol);
        hb_u32(out, &kRleMinLength, lay->rle_min_length);
        hps_write_uint_config(out, &kRunLengthConfig, 8);
    }
    const int ret = hps_write_cluster_map(lay->cluster_map, lay->num_dists, lay->num_clusters, out, err);
    if (ret) {
        if (err && !*err)
            *err = ret == ST_NOMEM ? "out of memory in prefix coder" : "prefix coder internal error";
        return ret;
    }
    hb_bool(out, 1); /* prefix codes */
    for (size_t c = 0; c < lay->num_clusters; c++)
        hps_write_uint_config(out, &lay->config[c], 15);
    return 0;
}

int hps_write_header(HydBits *out, const HydPrefixLayout *lay, const uint32_t *lengths, const char **err) {
    int ret = hps_write_header_fixed(out, lay, err);
    if (ret)
        return ret;

    for (size_t c = 0; c < lay->num_clusters; c++) {
        if (lay->alphabet[c] <= 1) {
            hb_bool(out, 0);
            continue;
        }
        hb_bool(out, 1);
        const int n = ilog2_u32(lay->alphabet[c] - 1u);
        hb_put(out, (uint64_t)n, 4);
        hb_put(out, lay->alphabet[c] - 1u, n);
    }

    size_t base = 0;
    for (size_t c = 0; c < lay->num_clusters; base += lay->alphabet[c], c++) {
        const uint32_t n = lay->alphabet[c];
        const uint32_t *len = lengths + base;
        if (n <= 1)
            continue;
        uint32_t used = 0;
        Pick pick[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
        for (uint32_t j = 0; j < n; j++) {
            if (!len[j])
                continue;
            if (used < 4) {
                pick[used].symbol = j;
                pick[used].len = len[j];
            }
            if (++used > 4)
                break;
        }
        if (used > 4) {
            ret = put_complex_lengths(out, n, len);
            if (ret)
                goto done;
            continue;
        }
        if (!used) {
            used = 1;
            pick[0].symbol = n - 1;
        }
        hb_put(out, 1, 2); /* hskip = 1: the "simple" form with up to four symbols */
        hb_put(out, used - 1, 2);
        /* the decoder assigns lengths by position: (1,2,2) for three symbols, (2,2,2,2) or
         * (1,2,3,3) for four, so the shortest code has to be listed first */
        if (used == 3 && pick[0].len != 1)
            swap_pick(&pick[0], pick[1].len == 1 ? &pick[1] : &pick[2]);
        int skewed = 0;
        if (used == 4) {
            for (int i = 0; i < 4; i++)
                if (pick[i].len != 2)
                    skewed = 1;
            if (skewed && pick[0].len != 1)
                swap_pick(&pick[0], pick[1].len == 1 ? &pick[1] : pick[2].len == 1 ? &pick[2] : &pick[3]);
            if (skewed && pick[1].len != 2)
                swap_pick(&pick[1], pick[2].len == 2 ? &pick[2] : &pick[3]);
        }
        const int symbol_bits = clog2_u32(n);
        for (uint32_t i = 0; i < used; i++)
            hb_put(out, pick[i].symbol, symbol_bits);
        if (used == 4)
            hb_bool(out, skewed);
    }
done:
    if (ret && err && !*err)
        *err = ret == ST_NOMEM ? "out of memory in prefix coder" : "prefix coder internal error";
    return ret;
}

int hps_finish_prefix(HydSymStream *s, HydBits *out, const char **err) {
    int ret = 0;
    uint32_t *freq = NULL, *lengths = NULL;
    Code *codes = NULL;
    size_t base[257];

    if (s->rle_min_symbol) {
        ret = flush_run(s);
        if (ret)
            goto done;
    }

    /* ---- histograms and code lengths ---- */
    base[0] = 0;
    for (size_t c = 0; c < s->num_clusters; c++)
        base[c + 1] = base[c] + s->alphabet[c];
    const size_t total = base[s->num_clusters] ? base[s->num_clusters] : 1;
    freq = calloc(total, sizeof(uint32_t));
    lengths = calloc(total, sizeof(uint32_t));
    codes = calloc(total, sizeof(Code));
    if (!freq || !lengths || !codes) {
        ret = ST_NOMEM;
        goto done;
    }
    for (size_t i = 0; i < s->count; i++)
        freq[base[s->sym[i].cluster] + s->sym[i].token]++;
    for (size_t c = 0; c < s->num_clusters; c++) {
        const uint32_t n = s->alphabet[c];
        if (n <= 1)
            continue;
        ret = hps_code_lengths(freq + base[c], lengths + base[c], n, 15);
        if (!ret)
            ret = assign_codes(codes + base[c], lengths + base[c], n);
        if (ret)
            goto done;
    }

    /* ---- header, then the symbols (entropy.c:1003-1021) ---- */
    {
        const HydPrefixLayout lay = {s->rle_min_symbol, s->rle_min_length, s->cluster_map, s->num_dists,
                                     s->num_clusters,   s->config,         s->alphabet};
        ret = hps_write_header(out, &lay, lengths, err);
        if (ret)
            goto done;
    }
    for (size_t i = 0; i < s->count; i++) {
        const HydSym *sym = &s->sym[i];
        const Code *code = &codes[base[sym->cluster] + sym->token];
        hb_put(out, code->bits, (int)code->len);
        hb_put(out, sym->residue, sym->residue_bits);
    }
    if (out->failed || s->failed)
        ret = ST_NOMEM;

done:
    if (ret && err && !*err)
        *err = ret == ST_NOMEM ? "out of memory in prefix coder" : "prefix coder internal error";
    free(freq);
    free(lengths);
    free(codes);
    hps_free(s);
    return ret;
}
