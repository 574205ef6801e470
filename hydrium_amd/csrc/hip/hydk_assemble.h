/*
 * hydk_assemble.h — private interface between the host-side planner (csrc/host/assembler.c, C99) and
 * the device-side frame assembler (csrc/hip/assemble.hip).  Nothing here is exported from the library.
 *
 * A PLAN is everything about a frame that does not depend on its pixels: the file and frame headers
 * (with the TOC permutation), LFGlobal, the constant bits in front of every LF group's coefficient
 * stream and the geometry-only bits behind it, the HFGlobal fields in front of the histograms, and the
 * order in which the LF groups arrive (which shard blob, which slot).  The planner builds it with the
 * host frame code (frame.c / prefix.c) once per frame shape; the assembler's kernels add what the
 * pixels decide — LF code headers, histograms, TOC sizes — and copy every section into place.
 */
#ifndef HYDK_ASSEMBLE_H_
#define HYDK_ASSEMBLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HYDK_ASM_MAX_BLOBS 64
#define HYDK_ASM_MAX_TAILS 4
#define HYDK_ASM_PLAN_MAGIC 0x4E4C5048u /* "HPLN" */

typedef struct HydkAsmSlot {   /* one LF group, in send order (= TOC order of the LF group sections) */
    uint32_t blob, index;      /* which blob, which slot record inside it */
    uint32_t preset;           /* raster id of the LF group = its HF preset: checked against the blob */
    uint32_t tail;             /* which geometry tail closes its section */
    uint32_t ngroups;          /* 256 x 256 groups it holds */
    uint32_t group_base;       /* groups of the LF groups sent before it */
    uint32_t pad[2];
} HydkAsmSlot;

typedef struct HydkAsmPlan {   /* header of the plan buffer; offsets are bytes from its start, 16-byte aligned */
    uint32_t magic, total_bytes;
    uint32_t num_slots, num_blobs, num_presets, clusters_per_preset;
    uint32_t toc_n, frame_groups;
    uint32_t prefix_off, prefix_bytes;       /* file header (optional) + frame header, byte aligned */
    uint32_t lfglobal_off, lfglobal_bytes;   /* the LFGlobal section */
    uint32_t lfpre_off, lfpre_bits;          /* LF group: modular header, MA tree and the fixed fields of the stream header */
    uint32_t hfpre_off, hfpre_bits;          /* HFGlobal up to and including "ANS, not prefix codes" */
    uint32_t ntails, tail_off[HYDK_ASM_MAX_TAILS], tail_bits[HYDK_ASM_MAX_TAILS];
    uint32_t slots_off;                      /* HydkAsmSlot[num_slots] */
    uint32_t preset_slot_off;                /* uint32[num_presets]: slot that carries preset p */
    uint32_t blob_slots[HYDK_ASM_MAX_BLOBS]; /* slot records each blob must hold */
    uint32_t blob_first[HYDK_ASM_MAX_BLOBS]; /* its first slot in send order */
} HydkAsmPlan;

typedef struct HydkAsm HydkAsm;

/* all return a HYDStatusCode-compatible value; hydk_asm_error() describes the last failure */
int hydk_asm_create(int device, HydkAsm **out);
void hydk_asm_destroy(HydkAsm *a);
const char *hydk_asm_error(HydkAsm *a);
/* copies the plan to the device (synchronous; once per frame shape) */
int hydk_asm_set_plan(HydkAsm *a, const void *plan, size_t bytes);
/* enqueue the assembly of one frame on `stream`: blobs are DEVICE pointers to hydamd_export_frame blobs
 * (complete once the work already in `stream` has run), `out` any device-accessible buffer of out_cap bytes */
int hydk_asm_run(HydkAsm *a, const void *const *blobs, const uint64_t *blob_caps, void *stream, void *out, uint64_t out_cap);
/* after the stream has been synchronised: bytes of the frame (0 on failure) and the device's error word */
int hydk_asm_result(HydkAsm *a, uint64_t *size, uint32_t *err);
/* ... and the frame itself, copied from where the last run wrote it (out == NULL in hydk_asm_run: the assembler's own buffer) */
int hydk_asm_read(HydkAsm *a, uint8_t *dst, size_t capacity);
/* debugging / tests: the scratch the kernels left (host copies; any pointer may be NULL) */
int hydk_asm_debug(HydkAsm *a, uint32_t slot, uint32_t *head_bits, uint32_t *head_words, size_t head_cap,
                   uint32_t *hfg_bits, uint32_t *hfg_words, size_t hfg_cap);

/* error word the kernels leave */
#define HYDK_ASM_E_BLOB 1u      /* malformed or unexpected blob header */
#define HYDK_ASM_E_RETRY 2u     /* a blob is incomplete (its frame outgrew a buffer) */
#define HYDK_ASM_E_NAN 4u       /* non-finite float sample */
#define HYDK_ASM_E_SLOT 8u      /* slot record inconsistent (preset, table or LF code error, LF stream out of range) */
#define HYDK_ASM_E_HEAD 16u     /* LF code header could not be built */
#define HYDK_ASM_E_SIZE 32u     /* section sizes inconsistent, or a section too large for the TOC */
#define HYDK_ASM_E_SPACE 64u    /* frame larger than the output buffer (result size = bytes needed) */
#define HYDK_ASM_E_SCRATCH 128u /* HFGlobal or TOC larger than the assembler's scratch */

#ifdef __cplusplus
}
#endif

#endif /* HYDK_ASSEMBLE_H_ */
