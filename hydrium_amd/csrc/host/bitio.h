/*
 * bitio.h — growable LSB-first bit sink used by the host-side frame assembler.
 *
 * Bit layout contract (reference src/libhydrium/bitwriter.c:110-124): a value is masked to its
 * width and its bit 0 lands at the current end of the stream; bytes are little-endian in bit
 * order.  Unlike the reference writer this one always owns its memory (the reference's can
 * realloc() a caller's buffer, SURVEY.md §8b "do not replicate").
 */
#ifndef HYD_BITIO_H_
#define HYD_BITIO_H_

#include <stddef.h>
#include <stdint.h>

typedef struct HydBits {
    uint8_t *data;   /* whole bytes emitted so far */
    size_t len;      /* number of whole bytes */
    size_t cap;
    uint64_t acc;    /* pending bits, bit 0 first */
    int nacc;        /* 0..63 */
    int failed;      /* sticky out-of-memory flag */
} HydBits;

/* JPEG XL U32() field: four (offset, extra-bit-count) alternatives selected by a 2-bit prefix. */
typedef struct HydU32Dist {
    uint32_t offset[4];
    uint32_t bits[4];
} HydU32Dist;

void hb_init(HydBits *b);
void hb_free(HydBits *b);
void hb_reset(HydBits *b);
/* append the low `nbits` (0..56) bits of value */
void hb_put(HydBits *b, uint64_t value, int nbits);
static inline void hb_bool(HydBits *b, int flag) { hb_put(b, flag ? 1 : 0, 1); }
/* zero bits up to the next byte boundary, then move all pending bits into data[] */
void hb_align(HydBits *b);
/* total bits written */
static inline uint64_t hb_bit_count(const HydBits *b) { return (uint64_t)b->len * 8 + (uint64_t)b->nacc; }
/* append `nbits` bits taken LSB-first from src (bit-granular splice, bitwriter.c:80-108) */
void hb_append_bits(HydBits *b, const uint8_t *src, uint64_t nbits);
/* grow a byte-aligned sink by n bytes the caller fills in; NULL if unaligned or out of memory
 * (the pointer is valid until the next call on this sink) */
uint8_t *hb_extend(HydBits *b, size_t n);
/* append whole bytes; the sink must be byte-aligned */
void hb_append_bytes(HydBits *b, const uint8_t *src, size_t n);

/* field encoders (bitwriter.c:134-196) */
int hb_u32(HydBits *b, const HydU32Dist *d, uint32_t value);
void hb_u64(HydBits *b, uint64_t value);
int hb_enum(HydBits *b, uint32_t value);
void hb_icc_varint(HydBits *b, uint64_t value);

#endif /* HYD_BITIO_H_ */
