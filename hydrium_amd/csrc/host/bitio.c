/* bitio.c — see bitio.h */
#include "bitio.h"

#include <stdlib.h>
#include <string.h>

void hb_init(HydBits *b) { memset(b, 0, sizeof(*b)); }

void hb_free(HydBits *b) {
    free(b->data);
    memset(b, 0, sizeof(*b));
}

void hb_reset(HydBits *b) {
    b->len = 0;
    b->acc = 0;
    b->nacc = 0;
}

static int hb_reserve(HydBits *b, size_t extra) {
    if (b->failed)
        return 0;
    if (b->len + extra <= b->cap)
        return 1;
    size_t ncap = b->cap ? b->cap : 4096;
    while (ncap < b->len + extra)
        ncap *= 2;
    uint8_t *nd = realloc(b->data, ncap);
    if (!nd) {
        b->failed = 1;
        return 0;
    }
    b->data = nd;
    b->cap = ncap;
    return 1;
}

static void hb_spill(HydBits *b) {
    if (!hb_reserve(b, 8))
        return;
    while (b->nacc >= 8) {
        b->data[b->len++] = (uint8_t)b->acc;
        b->acc >>= 8;
        b->nacc -= 8;
    }
}

void hb_put(HydBits *b, uint64_t value, int nbits) {
    if (nbits <= 0)
        return;
    if (nbits < 64)
        value &= (UINT64_C(1) << nbits) - 1;
    if (b->nacc + nbits > 64)
        hb_spill(b); /* leaves < 8 pending bits, and nbits <= 56 */
    b->acc |= value << b->nacc;
    b->nacc += nbits;
    if (b->nacc >= 32)
        hb_spill(b);
}

void hb_align(HydBits *b) {
    b->nacc = (b->nacc + 7) & ~7; /* the pad bits are already zero in acc */
    hb_spill(b);
}

void hb_append_bytes(HydBits *b, const uint8_t *src, size_t n) {
    if (b->nacc) { /* not aligned: fall back to the bit splice */
        hb_append_bits(b, src, (uint64_t)n * 8);
        return;
    }
    if (!n || !hb_reserve(b, n))
        return;
    memcpy(b->data + b->len, src, n);
    b->len += n;
}

uint8_t *hb_extend(HydBits *b, size_t n) {
    hb_spill(b);
    if (b->nacc || !hb_reserve(b, n ? n : 1))
        return NULL;
    uint8_t *p = b->data + b->len;
    b->len += n;
    return p;
}

void hb_append_bits(HydBits *b, const uint8_t *src, uint64_t nbits) {
    hb_spill(b);
    if (!b->nacc && nbits >= 8) {
        const size_t whole = (size_t)(nbits >> 3);
        hb_append_bytes(b, src, whole);
        src += whole;
        nbits &= 7;
    }
    while (nbits >= 32) { /* unaligned splice, four bytes at a time */
        hb_put(b, (uint64_t)src[0] | ((uint64_t)src[1] << 8) | ((uint64_t)src[2] << 16) | ((uint64_t)src[3] << 24), 32);
        src += 4;
        nbits -= 32;
    }
    while (nbits >= 8) {
        hb_put(b, *src++, 8);
        nbits -= 8;
    }
    if (nbits)
        hb_put(b, *src, (int)nbits);
}

int hb_u32(HydBits *b, const HydU32Dist *d, uint32_t value) {
    for (int i = 0; i < 4; i++) {
        const uint64_t span = d->bits[i] >= 64 ? ~UINT64_C(0) : (UINT64_C(1) << d->bits[i]) - 1;
        const uint64_t rel = (uint64_t)value - d->offset[i];
        if (value >= d->offset[i] && rel <= span) {
            hb_put(b, (rel << 2) | (uint64_t)i, (int)d->bits[i] + 2);
            return 0;
        }
    }
    return -1;
}

void hb_u64(HydBits *b, uint64_t value) {
    if (!value) {
        hb_put(b, 0, 2);
    } else if (value < 17) {
        hb_put(b, ((value - 1) << 2) | 1, 6);
    } else if (value < 273) {
        hb_put(b, ((value - 17) << 2) | 2, 10);
    } else {
        hb_put(b, ((value & 0xFFF) << 2) | 3, 14);
        for (int shift = 12;; shift += 8) {
            const uint64_t rest = value >> shift;
            if (!rest) {
                hb_put(b, 0, 1);
                return;
            }
            if (shift == 60) {
                hb_put(b, ((rest & 0xF) << 1) | 1, 5);
                return;
            }
            hb_put(b, ((rest & 0xFF) << 1) | 1, 9);
        }
    }
}

int hb_enum(HydBits *b, uint32_t value) {
    static const HydU32Dist dist = {{0, 1, 2, 18}, {0, 0, 4, 6}};
    if (value > 63)
        return -1;
    return hb_u32(b, &dist, value);
}

void hb_icc_varint(HydBits *b, uint64_t value) {
    while (value > 0x7f) {
        hb_put(b, (value & 0x7f) | 0x80, 8);
        value >>= 7;
    }
    hb_put(b, value & 0x7f, 8);
}
