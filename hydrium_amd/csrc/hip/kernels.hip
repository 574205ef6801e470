/*
 * kernels.hip — hand-written CDNA4 (gfx950) kernels for hydrium's per-group encode hot path.
 *
 *   k_transform_tokenize  one 256x256 group per workgroup, strip-sequential (8 px rows at a time):
 *                         RGB -> XYB (reference format.c:15-56,85-140), 8x8 forward DCT
 *                         (encoder.c:631-668), HF quantisation + LF ints (encoder.c:573-582,
 *                         783-823), tokenisation into hybrid-uint symbols (encoder.c:689-750,
 *                         entropy.c:427-444) and the per-preset token histogram (entropy.c:526-544).
 *   k_build_tables        histogram -> 12-bit frequencies -> alias table -> inverse slot table
 *                         (entropy.c:184-301, 943-978).
 *   k_rans_encode         one wave per group: the serial reverse rANS chain (entropy.c:1064-1159)
 *                         with wave-parallel bit emission, written back-to-front so that no
 *                         replay pass is needed (lowest latency of one frame; float input).
 *   k_rans_lanes / k_rans_emit   the same chain with one LANE per group — a wavefront walks the 64
 *                         chains of an LF group and records each step's refill word — and a
 *                         wave-parallel kernel that writes the bits straight into the frame's
 *                         payload (highest frame rate: 0.3 instructions per symbol).
 *   k_scan_sections / k_pack_sections   byte sizes and offsets of the HF sections; packing of the wave form's.
 *
 * Arithmetic contract: IEEE binary32, source operation order, NO fused multiply-add — the
 * reference's canonical bytes are the non-contracted ones.  This file is
 * compiled with -ffp-contract=off and additionally pins the pragma below.  The 8-point DCT is an
 * ordered 8-term accumulation, so it runs on the VALU (MFMA would fuse and re-associate).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hydk_common.h"
#include <atomic>
#include <type_traits>

#pragma clang fp contract(off)

namespace {

constexpr int kThreads = 256;
#ifndef HYDK_K1_WAVES
#define HYDK_K1_WAVES 4 /* waves per SIMD the transform kernel is compiled for (register budget 512 / this) */
#endif
/* Build-time variants of the transform kernel, for A/B measurements (scripts/k1_variants.py; results in
 * profiles/r04_k1_variants.txt and DESIGN.md 3, 9); the defaults are the product.
 *   HYDK_K1_GATHER     bit i set = LUT i of a pixel is a gather from the uploaded table instead of a register evaluation
 *                      (0-2 transfer curve of R, G, B; 3-5 bias curve of L, M, S).  Default 0x10: the M channel's bias curve
 *                      through the otherwise idle texture path — photo 0.467 -> 0.424 ms alone, smooth 0.423 -> 0.390,
 *                      RGB8 photo 0.396 -> 0.375, the pipelined loop +0.7 % (146.7 -> 147.7 Gpixel/s); random-noise pixels,
 *                      whose indices scatter over the whole 256 KB table, 1.04 -> 1.17.  More gathers cost more than they save.
 *   HYDK_K1_ILP        pixels of a row whose curves are evaluated in lock step (0: one value at a time, as until round 3)
 *   HYDK_K1_WAVELOCAL  a wavefront row-transforms exactly the eight blocks whose columns it transforms next, so no
 *                      workgroup barrier separates the two phases
 *   HYDK_K1_SKIP       timing-only builds that leave a stage out (wrong bytes): 1 token walk, 2 curves, 4 bitmaps, 8 column pass
 * (Round 4's occupancy switches — HYDK_K1_PADLDS, HYDK_K1_WAVES_EXACT, HYDK_K1_NUM_VGPR, HYDK_LANES_PRIO — and the per-phase
 * cycle counters are gone with the experiments they served: profiles/r04_pipeline_bounds.txt) */
#ifndef HYDK_K1_GATHER
#define HYDK_K1_GATHER 0x10
#endif
#ifndef HYDK_K1_WAVELOCAL
#define HYDK_K1_WAVELOCAL 1
#endif
#ifndef HYDK_K1_SKIP
#define HYDK_K1_SKIP 0
#endif
#ifndef HYDK_K1_ILP
#define HYDK_K1_ILP 2
#endif
/*   HYDK_K1_TRIM       round 6: the token walk's 64-bit multiply-add (cluster * 40 + token) as a 24-bit one, and the record
 *                      store's address as scalar base + 32-bit offset */
#ifndef HYDK_K1_TRIM
#define HYDK_K1_TRIM 0
#endif
/*   HYDK_K1_CHANSEQ    round 6: the three channels take the row-pass buffer IN TURN (row DCT of channel c -> LDS -> column DCT,
 *                      quantiser of channel c, then the next channel; a wavefront reads only what it wrote itself, no barrier)
 *                      and the quantised coefficients live as 16-bit values in an array of their own, a thread's eight in one
 *                      16-byte store ([block][kh][kv]): 9.0 + 12.0 KB of LDS where all three channels' row passes took 27.0 —
 *                      25 712 bytes = 21 granules per workgroup instead of 25, so that THREE transform workgroups fit beside a
 *                      chain workgroup's 63 granules instead of two (profiles/r06_nc_probe.txt: that occupancy is worth +4 %) */
#ifndef HYDK_K1_CHANSEQ
#define HYDK_K1_CHANSEQ 0
#endif
/* Round 6 (VERDICT r5 task 1): what of a lane-form chain's work costs the pipelined loop?  Timing-only variants of the
 * chain kernel (wrong bytes; scripts/k1_variants.py builds them, scripts/pipe_probe.py runs them with the emit stage off):
 *   HYDK_CHAIN_PROBE   1: every operand row from ONE address (no bank conflicts among the 64 lanes' ds_read_b128);
 *                      4: no global traffic after the first round (a lane walks its first 16 records again and again and
 *                         stores nothing); 8: no stores only; 16: no loads only; 32: every chain wavefront leaves its start and end time
 *                         where the section sizes go (scripts/pipe_probe.py --chain-clock)
 *   HYDK_CHAIN_PRIO    issue priority of the chain wavefronts (product: 3)
 *   HYDK_K1_PRIO       issue priority of the transform kernel's wavefronts (product: none set = 0) */
#ifndef HYDK_CHAIN_PROBE
#define HYDK_CHAIN_PROBE 0
#endif
/*   HYDK_LANE_STEP     the lane-form chain's step: 1 = round 5's (the renormalised state kept as two pieces B | slot & mask:
 *                      13.5 vector instructions per symbol, one of them between the slot's arrival and the multiply);
 *                      2 = round 6's (the refill decision carried in a scalar register pair, x = decision ? state >> 16 :
 *                      state by ONE v_cndmask_sdwa: 11.5 vector instructions per symbol, two between arrival and multiply).
 *                      What the chains cost the pipelined loop is the vector issue time they take from the ONE SIMD they
 *                      share with transform wavefronts, times four (a transform workgroup spans the four SIMDs and moves
 *                      at its slowest wavefront's pace; profiles/r06_chain_probes.txt): fewer instructions, not a shorter
 *                      dependent path, is what the loop pays for. */
#ifndef HYDK_LANE_STEP
#define HYDK_LANE_STEP 1
#endif
/*   HYDK_LANE_PIPE     how the lane-form chain's record lines travel: 0 = round 5's (one buffer copied into another at every
 *                      round's start, the line one round ahead, stores in the middle of the walk); 1 / 2 = two / three
 *                      buffers taking turns, the line 1 / 2 rounds ahead, stores at the round's start (see HYDK_LANE_ROUND) */
#ifndef HYDK_LANE_PIPE
#define HYDK_LANE_PIPE 2
#endif
#ifndef HYDK_CHAIN_PRIO
#define HYDK_CHAIN_PRIO 3
#endif
/*   HYDK_LANE_TAB_GLOBAL  1: the lane-form chain looks its slots up in the table kernel's output where it lies (global memory,
 *                      L2-resident: 72 KB per LF group) instead of a copy in LDS: the workgroup holds 6 KB of LDS (operand rows)
 *                      where it held 80, a step waits for an L2 round trip where it waited for LDS.  Round 6: the stand-in
 *                      that issues a chain's instructions and holds NO LDS costs the pipelined loop 8 % where the chains
 *                      cost 20 (profiles/r06_chain_probes.txt) — what is that worth when the step is the real one, 2-3 x slower? */
#ifndef HYDK_LANE_TAB_GLOBAL
#define HYDK_LANE_TAB_GLOBAL 0
#endif
/*   HYDK_LANE_NC9_PROBE  timing only (wrong bytes; run with the emit stage off): a nine-cluster frame's chains run the instance
 *                      that holds tables for this many clusters — 7: 61.8 KB, 6: 53 KB, 4: 35 KB instead of 79.5 — : what would a
 *                      chain be worth beside which THREE transform workgroups fit (160 KB - 3 x 31.25 = 66 KB)? */
/*   (The compiler PADS the register allocation of a kernel whose STATIC LDS lets only two of its workgroups onto a compute unit —
 *   such a kernel can never have more than one wavefront per SIMD — to the smallest figure that guarantees it:
 *   .amdhsa_next_free_vgpr 257 for the chain kernel, which uses 164.  amdgpu_waves_per_eu(1, N) and amdgpu_num_vgpr(N) do not move
 *   it; asking for the LDS at launch does: HYDK_CHAIN_DYN_LDS.) */
#ifndef HYDK_CHAIN_LDS_MIN
#define HYDK_CHAIN_LDS_MIN 0
#endif
#ifndef HYDK_LANE_NC9_PROBE
#define HYDK_LANE_NC9_PROBE 9
#endif
/*   HYDK_CHAIN_HOG     1: a chain wavefront names accumulation register a255, so that it is allocated 256 of them on top of its
 *                      vector registers and no transform wavefront (120) fits beside it on its SIMD: the chain keeps its
 *                      SIMD's issue port to itself, the transform workgroups of its compute unit live on the other three */
#ifndef HYDK_CHAIN_HOG
#define HYDK_CHAIN_HOG 0
#endif
#ifndef HYDK_K1_PRIO
#define HYDK_K1_PRIO 0
#endif
constexpr int kS0Block = 72;            /* floats per block in the row-pass buffer: 64 + 8 pad -> conflict-free column reads */
constexpr int kS0Chan = 32 * kS0Block;  /* floats per channel */
constexpr int kDbgPitch = 2048;

/* |cos| magnitudes of the scaled DCT-II as the reference spells them (encoder.c:32-40):
 * double literals narrowed to float at compile time. */
constexpr float kA = (float)0.17338, kB = (float)0.146984, kC = (float)0.0982119, kD = (float)0.0344874;
constexpr float kE = (float)0.16332, kF = (float)0.0676495, kG = (float)0.125;

constexpr float kDct[7][8] = {
    {kA, kB, kC, kD, -kD, -kC, -kB, -kA},
    {kE, kF, -kF, -kE, -kE, -kF, kF, kE},
    {kB, -kD, -kA, -kC, kC, kA, kD, -kB},
    {kG, -kG, -kG, kG, kG, -kG, -kG, kG},
    {kC, -kA, kD, kB, -kB, -kD, kA, -kC},
    {kF, -kE, kE, -kF, -kF, kE, -kE, kF},
    {kD, -kC, kB, -kA, kA, -kB, kC, -kD},
};

/* zig-zag index of the coefficient with vertical frequency kv and horizontal frequency kh: the
 * reference's natural_order[j] = {x = kv, y = kh} (encoder.c:42-51 with the transposed store of
 * encoder.c:660-664). */
__device__ const uint8_t kZigzag[8][8] = {
    /* kv = 0 */ {0, 2, 3, 9, 10, 20, 21, 35},
    /* kv = 1 */ {1, 4, 8, 11, 19, 22, 34, 36},
    /* kv = 2 */ {5, 7, 12, 18, 23, 33, 37, 48},
    /* kv = 3 */ {6, 13, 17, 24, 32, 38, 47, 49},
    /* kv = 4 */ {14, 16, 25, 31, 39, 46, 50, 57},
    /* kv = 5 */ {15, 26, 30, 40, 45, 51, 56, 58},
    /* kv = 6 */ {27, 29, 41, 44, 52, 55, 59, 62},
    /* kv = 7 */ {28, 42, 43, 53, 54, 60, 61, 63},
};

/* [kh][kv nibble][4 non-zero flags] -> OR of the zig-zag bits of coefficients (4 * nibble + b, kh), b in the flags:
 * turns a thread's eight "non-zero" flags into its share of the block's zig-zag bitmap with two loads */
struct NibbleMasks {
    unsigned long long m[8][2][16];
    constexpr NibbleMasks() : m() {
        constexpr uint8_t zz[8][8] = {{0, 2, 3, 9, 10, 20, 21, 35},   {1, 4, 8, 11, 19, 22, 34, 36},  {5, 7, 12, 18, 23, 33, 37, 48},
                                      {6, 13, 17, 24, 32, 38, 47, 49}, {14, 16, 25, 31, 39, 46, 50, 57}, {15, 26, 30, 40, 45, 51, 56, 58},
                                      {27, 29, 41, 44, 52, 55, 59, 62}, {28, 42, 43, 53, 54, 60, 61, 63}};
        for (int kh = 0; kh < 8; kh++)
            for (int half = 0; half < 2; half++)
                for (int pat = 0; pat < 16; pat++) {
                    unsigned long long v = 0;
                    for (int b = 0; b < 4; b++)
                        if (pat >> b & 1)
                            v |= 1ull << zz[4 * half + b][kh];
                    m[kh][half][pat] = v;
                }
    }
};
__device__ const NibbleMasks kNibbleMasks = NibbleMasks();

/* HF quantisation weights, channel X / Y / B, zig-zag order (encoder.c:74-93) */
__device__ const int16_t kQuantWeight[3][64] = {
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

/* coefficient-count context offsets (encoder.c:60-66); all values fit a byte */
__device__ const uint8_t kNnzCtx[64] = {
    0,   0,   31,  62,  62,  93,  93,  93,  93,  123, 123, 123, 123, 152, 152, 152, 152, 152, 152, 152, 152, 180,
    180, 180, 180, 180, 180, 180, 180, 180, 180, 180, 180, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206,
    206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206,
};

constexpr float kLfShift[3] = {8192.f, 1024.f, 512.f}; /* encoder.c:573 */

/* ------------------------------------------------------------------------------------------
 * pixel front-end
 * ---------------------------------------------------------------------------------------- */

constexpr float kUnit16 = 1.0f / (65536 - 1.0f); /* format.c:64,79: 1.0f / (size - 1.0f) */

__device__ __forceinline__ float linearize(float x) { /* format.c:15-19 */
    if (x <= 0.0404482362771082f)
        return 0.07739938080495357f * x;
    return 0.003094300919832f + x * (-0.009982599f + x * (0.72007737769f + 0.2852804880f * x));
}

__device__ __forceinline__ float bias_curve(float v) { /* format.c:21-31 */
    const float x = v + 0.0037930732552754493f;
    float z = __uint_as_float(0x548c39cbu - __float_as_uint(x) / 3u);
    z *= 1.5015480449f - 0.534850249f * x * z * z * z;
    z *= 1.333333985f - 0.33333333f * x * z * z * z;
    return 1.0f / z - 0.155954f; /* correctly rounded division (hipcc default) */
}

__device__ __forceinline__ uint32_t to_u16(float x) { /* format.c:33-36 */
    const int y = (int)(x * 65535.f + 0.5f);
    return (uint32_t)(y < 0 ? 0 : y > 65535 ? 65535 : y);
}

/* XYB evaluation modes of the integer pixel path (the launcher picks the first one whose
 * register evaluation reproduces all 65536 host-built LUT entries bit for bit):
 *   0  registers, reciprocal by v_rcp_f32 + one fused Newton step
 *   1  registers, IEEE division as the compiler expands it
 *   2  gathers from the uploaded LUTs */
constexpr int kXybFastRcp = 0, kXybIeeeDiv = 1, kXybGather = 2;
/* 3, 4: modes 0 and 1 with EVERY curve evaluated in registers — no HYDK_K1_GATHER gathers.  Chosen per frame by the host
 * (device_api.hip transform_range) for content whose pixels scatter over the whole bias table, where a gather is an L2
 * miss per lane (random noise: 1.04 ms against 1.17 with the M-channel gather) */
constexpr int kXybFastRcpRegs = 3, kXybIeeeDivRegs = 4;
constexpr int xyb_arith(int xm) { return xm == kXybFastRcpRegs ? kXybFastRcp : xm == kXybIeeeDivRegs ? kXybIeeeDiv : xm; }
constexpr int xyb_gmask(int xm) { return xm == kXybFastRcpRegs || xm == kXybIeeeDivRegs ? 0 : HYDK_K1_GATHER; }

/* the 65536-entry LUTs of format.c:58-83, evaluated in registers.  The clamp of f32_to_u16 (format.c:33-36)
 * never acts on these inputs — the curve maps [0, 1] into [0, 1) — so it is left out; like every other
 * shortcut here that is not assumed but checked entry by entry by k_lut_selftest. */
__device__ __forceinline__ uint32_t input_lut16_eval(uint32_t i, int linear_light) {
    const float f = (float)i * kUnit16;
    return (uint32_t)(int)((linear_light ? f : linearize(f)) * 65535.f + 0.5f);
}
/* The same entry with the transfer-curve decision taken outside: samples above kDarkMax are on the cubic
 * branch of linearize (format.c:15-19), so a wavefront that holds none at or below it evaluates the cubic
 * alone — no compare, no masked second branch — and a linear-light job no curve at all.  That the branch
 * flips exactly between kDarkMax and kDarkMax + 1 is checked for all 65536 inputs by k_lut_selftest. */
constexpr uint32_t kDarkMax = 2650;
constexpr int kCurveBoth = 0, kCurveNone = 1, kCurveCubic = 2;
template <int CURVE>
__device__ __forceinline__ uint32_t input_lut16_eval_as(uint32_t i) {
    const float f = (float)i * kUnit16;
    const float y = CURVE == kCurveNone ? f : CURVE == kCurveCubic
        ? 0.003094300919832f + f * (-0.009982599f + f * (0.72007737769f + 0.2852804880f * f)) : linearize(f);
    return (uint32_t)(int)(y * 65535.f + 0.5f);
}
/* floor(u / 3) for u < 2^31 in one multiply-high: u * ceil(2^32 / 3) overshoots u / 3 by less than 1/3 */
__device__ __forceinline__ uint32_t div3(uint32_t u) { return __umulhi(u, 0x55555556u); }
template <int XMODE>
__device__ __forceinline__ float bias_lut_eval(uint32_t i) {
    if (xyb_arith(XMODE) == kXybIeeeDiv)
        return bias_curve((float)i * kUnit16);
    /* same operations as bias_curve (format.c:21-31) with 1.0f / z computed as v_rcp_f32 plus one
     * fused Newton step: explicit fmaf() is a fused operation regardless of the contraction setting,
     * and is used for the division only, whose IEEE result does not depend on how it is reached.
     * That one step reproduces the IEEE quotient for every z this LUT can see is not assumed but
     * checked: k_lut_selftest compares all 65536 entries at context creation, and a device where
     * it fails runs the IEEE-division variant instead. */
    const float x = (float)i * kUnit16 + 0.0037930732552754493f;
    float z = __uint_as_float(0x548c39cbu - div3(__float_as_uint(x))); /* x is a positive float below 2: its bits are below 2^30 */
    z *= 1.5015480449f - 0.534850249f * x * z * z * z;
    z *= 1.333333985f - 0.33333333f * x * z * z * z;
    const float r0 = __builtin_amdgcn_rcpf(z);
    const float r1 = __builtin_fmaf(r0, __builtin_fmaf(-z, r0, 1.0f), r0);
    return r1 - 0.155954f;
}

/* Pointers read from a job descriptor in memory are generic to the compiler (flat_* instructions, which also
 * count against the LDS counter every LDS wait looks at); all of them are device memory. */
#define HYDK_GLOBAL(T, p) ((__attribute__((address_space(1))) T *)(p))
/* The same two curves for N values in lock step: every statement is N independent operations, so a wavefront always
 * has N instructions to issue while the previous N are in the pipeline (a dependent VALU instruction issues ~9 cycles
 * after its producer, an independent one after ~2: the one-value forms above are chains of up to 21 dependent steps).
 * Operation order per value is exactly that of input_lut16_eval_as / bias_lut_eval. */
template <int CURVE, int N>
__device__ __forceinline__ void input_lut16_eval_n(const uint32_t (&i)[N], uint32_t (&o)[N]) {
    float f[N], y[N];
#pragma unroll
    for (int k = 0; k < N; k++)
        f[k] = (float)i[k] * kUnit16;
    if (CURVE == kCurveNone) {
#pragma unroll
        for (int k = 0; k < N; k++)
            y[k] = f[k];
    } else {
#pragma unroll
        for (int k = 0; k < N; k++)
            y[k] = 0.2852804880f * f[k];
#pragma unroll
        for (int k = 0; k < N; k++)
            y[k] = 0.72007737769f + y[k];
#pragma unroll
        for (int k = 0; k < N; k++)
            y[k] = f[k] * y[k];
#pragma unroll
        for (int k = 0; k < N; k++)
            y[k] = -0.009982599f + y[k];
#pragma unroll
        for (int k = 0; k < N; k++)
            y[k] = f[k] * y[k];
#pragma unroll
        for (int k = 0; k < N; k++)
            y[k] = 0.003094300919832f + y[k];
        if (CURVE == kCurveBoth) {
#pragma unroll
            for (int k = 0; k < N; k++)
                y[k] = f[k] <= 0.0404482362771082f ? 0.07739938080495357f * f[k] : y[k];
        }
    }
#pragma unroll
    for (int k = 0; k < N; k++)
        y[k] = y[k] * 65535.f;
#pragma unroll
    for (int k = 0; k < N; k++)
        y[k] = y[k] + 0.5f;
#pragma unroll
    for (int k = 0; k < N; k++)
        o[k] = (uint32_t)(int)y[k];
}

template <int XMODE, int N>
__device__ __forceinline__ void bias_lut_eval_n(const uint32_t (&i)[N], float (&o)[N]) {
    if (xyb_arith(XMODE) == kXybIeeeDiv) {
#pragma unroll
        for (int k = 0; k < N; k++)
            o[k] = bias_curve((float)i[k] * kUnit16);
        return;
    }
    float x[N], z[N], t[N];
#pragma unroll
    for (int k = 0; k < N; k++)
        x[k] = (float)i[k] * kUnit16;
#pragma unroll
    for (int k = 0; k < N; k++)
        x[k] = x[k] + 0.0037930732552754493f;
#pragma unroll
    for (int k = 0; k < N; k++)
        z[k] = __uint_as_float(0x548c39cbu - div3(__float_as_uint(x[k])));
    /* z *= 1.5015480449f - 0.534850249f * x * z * z * z */
#pragma unroll
    for (int k = 0; k < N; k++)
        t[k] = 0.534850249f * x[k];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int k = 0; k < N; k++)
            t[k] = t[k] * z[k];
#pragma unroll
    for (int k = 0; k < N; k++)
        t[k] = 1.5015480449f - t[k];
#pragma unroll
    for (int k = 0; k < N; k++)
        z[k] = z[k] * t[k];
    /* z *= 1.333333985f - 0.33333333f * x * z * z * z */
#pragma unroll
    for (int k = 0; k < N; k++)
        t[k] = 0.33333333f * x[k];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int k = 0; k < N; k++)
            t[k] = t[k] * z[k];
#pragma unroll
    for (int k = 0; k < N; k++)
        t[k] = 1.333333985f - t[k];
#pragma unroll
    for (int k = 0; k < N; k++)
        z[k] = z[k] * t[k];
#pragma unroll
    for (int k = 0; k < N; k++)
        t[k] = __builtin_amdgcn_rcpf(z[k]);
#pragma unroll
    for (int k = 0; k < N; k++)
        x[k] = __builtin_fmaf(-z[k], t[k], 1.0f);
#pragma unroll
    for (int k = 0; k < N; k++)
        t[k] = __builtin_fmaf(t[k], x[k], t[k]);
#pragma unroll
    for (int k = 0; k < N; k++)
        o[k] = t[k] - 0.155954f;
}

/* c * x + acc for c, x below 2^24, as the one instruction it is (the compiler keeps multiply and add apart) */
__device__ __forceinline__ uint32_t umad24(uint32_t c, uint32_t x, uint32_t acc) {
    uint32_t d;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "s"(c), "v"(x), "v"(acc));
    return d;
}

template <int XMODE>
__device__ __forceinline__ void lms_mix_u16(uint32_t r, uint32_t g, uint32_t b, const float *bias_lut, float &X,
                                            float &Y, float &B) {
    constexpr bool LUTS = XMODE == kXybGather;
    /* format.c:48-56: 16.16 fixed-point LMS mix, high half indexes the bias LUT */
    /* r, g, b are 16-bit LUT outputs: 24-bit multiplies are exact on them and fuse with the additions
     * (v_mad_u32_u24), same sums modulo 2^32 as the reference's 32-bit arithmetic */
    const uint32_t bb = __umul24(5112u, b);
    const uint32_t il = umad24(19661u, r, umad24(40761u, g, bb)) >> 16;
    const uint32_t im = umad24(15073u, r, umad24(45350u, g, bb)) >> 16;
    const uint32_t is = umad24(15953u, r, umad24(13419u, g, __umul24(36163u, b))) >> 16;
    float l, m, s;
    if (LUTS) {
        l = bias_lut[il];
        m = bias_lut[im];
        s = bias_lut[is];
    } else {
        const auto *gl = HYDK_GLOBAL(const float, bias_lut);
        l = (xyb_gmask(XMODE) & 8) ? gl[il] : bias_lut_eval<XMODE>(il);
        m = (xyb_gmask(XMODE) & 16) ? gl[im] : bias_lut_eval<XMODE>(im);
        s = (xyb_gmask(XMODE) & 32) ? gl[is] : bias_lut_eval<XMODE>(is);
    }
    Y = (l + m) * 0.5f;
    X = Y - m;
    B = s - Y;
}

__device__ __forceinline__ bool lms_mix_f32(float r, float g, float b, int linear_light, float &X, float &Y,
                                            float &B) {
    /* format.c:111-140, 38-46 */
    const bool finite = ((__float_as_uint(r) & 0x7f800000u) != 0x7f800000u) &&
                        ((__float_as_uint(g) & 0x7f800000u) != 0x7f800000u) &&
                        ((__float_as_uint(b) & 0x7f800000u) != 0x7f800000u);
    if (!linear_light) {
        r = linearize(r);
        g = linearize(g);
        b = linearize(b);
    }
    const float l = bias_curve(0.3f * r + 0.622f * g + 0.078f * b);
    const float m = bias_curve(0.23f * r + 0.692f * g + 0.078f * b);
    const float s = bias_curve(0.243423f * r + 0.204767f * g + 0.55181f * b);
    Y = (l + m) * 0.5f;
    X = Y - m;
    B = s - Y;
    return finite;
}

/* ------------------------------------------------------------------------------------------
 * 8-point DCT in the reference's summation order (encoder.c:639-658)
 * ---------------------------------------------------------------------------------------- */
__device__ __forceinline__ void dct8(const float (&x)[8], float (&o)[8]) {
    float dc = x[0];
#pragma unroll
    for (int n = 1; n < 8; n++)
        dc += x[n];
    o[0] = dc * 0.125f;
#pragma unroll
    for (int k = 1; k < 8; k++) {
        if (k == 4) {
            /* every coefficient of this row is +-0.125: multiplying by a power of two is exact and commutes
             * with the rounding of each addition (no intermediate is anywhere near the subnormal range:
             * samples are multiples of 2^-27 or row-pass outputs of such), so the eight products and seven
             * ordered additions collapse to seven ordered additions and one product — same bits */
            float acc = x[0];
#pragma unroll
            for (int n = 1; n < 8; n++)
                acc = kDct[3][n] > 0 ? acc + x[n] : acc - x[n];
            o[k] = acc * 0.125f;
            continue;
        }
        /* the reference starts from +0.0f; 0.0f + p differs from p only in the sign of a zero,
         * which no later stage can observe (every consumer multiplies and truncates to int) */
        float acc = x[0] * kDct[k - 1][0];
#pragma unroll
        for (int n = 1; n < 8; n++)
            acc += x[n] * kDct[k - 1][n];
        o[k] = acc;
    }
}

__device__ __forceinline__ uint32_t pack_signed(int v) { /* math-functions.h:68-71 */
    const uint32_t w = (uint32_t)v;
    return (w << 1) ^ (0u - (w >> 31));
}

template <int FMT>
struct SampleOf;
template <>
struct SampleOf<HYDK_FMT_U8> {
    typedef uint8_t type;
};
template <>
struct SampleOf<HYDK_FMT_U16> {
    typedef uint16_t type;
};
template <>
struct SampleOf<HYDK_FMT_F32> {
    typedef float type;
};

} /* namespace */

/* ==========================================================================================
 * K1: fused transform + tokenise.  grid = 64 group slots per LF group x LF groups of the frame,
 * block = 256 threads (4 waves); one 256x256 group per workgroup, walked as 32 strips of 8 rows.
 *
 * Per strip (32 varblocks x 3 channels):
 *   A  thread (row r, block b): 8 pixels -> XYB in registers -> three 8-point row DCTs -> LDS
 *   B  thread (block cb, horizontal frequency kh): three 8-point column DCTs, quantisation, the
 *      quantised coefficients back into the thread's own LDS column; 8-lane OR gives the block's
 *      non-zero bitmap; one thread per block leaves a descriptor per (block, channel) in LDS
 *   C1 every wave prefix-sums the 32 per-block symbol counts
 *   C2 the strip's symbol stream is cut into 256 equal runs; a thread finds its run's first
 *      (block, channel, zig-zag position) by binary search and walks it, carrying the contexts'
 *      state from symbol to symbol
 * Token order inside a group is block raster, channels Y, X, B (encoder.c:707-745), which the
 * strip order + prefix sum reproduces.  The next strip's pixels are in flight during B and C.
 * ======================================================================================== */
/* Everything but the transform kernel is a small amount of work on the critical path of a frame that is
 * already late (its stream holds nothing else): those kernels raise their wavefronts' issue priority so
 * that, sharing a SIMD with other frames' transform waves, they run as fast as they do alone. */
#define HYDK_URGENT() __builtin_amdgcn_s_setprio(3)

/* inclusive prefix sums in registers (DPP), no LDS round trips */
#define HYDK_DPP(v, ctrl, rmask) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), (rmask), 0xF, false))
__device__ __forceinline__ uint32_t scan16_inclusive(uint32_t v) { /* within each 16-lane row */
    v += HYDK_DPP(v, 0x111, 0xF); /* row_shr:1 */
    v += HYDK_DPP(v, 0x112, 0xF); /* row_shr:2 */
    v += HYDK_DPP(v, 0x114, 0xF); /* row_shr:4 */
    v += HYDK_DPP(v, 0x118, 0xF); /* row_shr:8 */
    return v;
}
__device__ __forceinline__ uint32_t scan64_inclusive(uint32_t v) {
    v = scan16_inclusive(v);
    v += HYDK_DPP(v, 0x142, 0xA); /* row_bcast:15 into rows 1 and 3 */
    v += HYDK_DPP(v, 0x143, 0xC); /* row_bcast:31 into rows 2 and 3 */
    return v;
}

/* OR of a 32-bit value over the 8 lanes of a varblock's thread group, in three DPP steps: the two
 * quad permutes complete each quad, row_half_mirror (lane i <-> 7 - i inside each half row) then
 * pairs every lane with one of the other quad */
__device__ __forceinline__ uint32_t or_reduce8(uint32_t v) {
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);  /* quad_perm:[1,0,3,2] */
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);  /* quad_perm:[2,3,0,1] */
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false); /* row_half_mirror */
    return v;
}

/* Token records.  Integer input: |quantised coefficient| <= 0.42 * 1969 * 5 < 2^13 (XYB is bounded by the
 * bias LUT's range and the scaled DCT has unit gain), so token < 36, residue < 2^13 and one record fits
 * 32 bits: residue bit count | symbol << 4 | residue << 16 (hydk_common.h), symbol = cluster * 40 + token — the histogram bin,
 * which the caller has at hand.  Float input has no such bound and keeps the 8-byte record:
 * lo = token | cluster << 8 | bit count << 16, hi = residue. */
template <int FMT>
__device__ __forceinline__ void store_record(void *tok, uint32_t at, uint32_t token, uint32_t cluster, uint32_t symbol,
                                             uint32_t rbits, uint32_t residue) {
    if (FMT == HYDK_FMT_F32)
        ((uint64_t *)tok)[at] = ((uint64_t)residue << 32) | HYDK_REC_LO(token, cluster, rbits);
    else {
#if HYDK_K1_TRIM
        /* a 32-bit byte offset from the (wave-uniform) array base: the store takes the base from scalar registers and the
         * offset from one vector register, no 64-bit address arithmetic per symbol */
        const uint32_t off = at * 4u;
        *(HYDK_GLOBAL(uint32_t, (char *)tok + off)) = HYDK_REC32(symbol, rbits, residue);
#else
        HYDK_GLOBAL(uint32_t, tok)[at] = HYDK_REC32(symbol, rbits, residue);
#endif
    }
}

#define HYDK_K1_OCCUPANCY __launch_bounds__(kThreads, HYDK_K1_WAVES)
template <int FMT, int XMODE>
__global__ HYDK_K1_OCCUPANCY void k_transform_tokenize(const HydkLfJob *__restrict__ jobs, uint32_t *status, uint2 *part_info,
                                                       int plog) {
    typedef typename SampleOf<FMT>::type sample_t;
    constexpr bool LUTS = XMODE == kXybGather;
    constexpr int kWords = FMT == HYDK_FMT_U8 ? 6 : 12; /* dwords holding 8 packed RGB pixels */
#if HYDK_K1_PRIO
    __builtin_amdgcn_s_setprio(HYDK_K1_PRIO);
#endif
    /* plog > 0 (launches of one or two LF groups: a tile-mode frame, the drop-in API's closing tile): a group is walked by
     * 1 << plog workgroups, a run of strips each, so that 64 or 128 groups still fill 256 compute units; every part leaves
     * its symbols in its own share of the group's token array and k_join_parts closes the gaps (round 5) */
    const unsigned bid = blockIdx.x >> plog, part = blockIdx.x & ((1u << plog) - 1u);
    const HydkLfJob job = jobs[bid >> 6];
    if (job.fmt != FMT || (FMT != HYDK_FMT_F32 && job.use_luts != xyb_arith(XMODE)))
        return; /* another template instance of this launch round owns this LF group */
    if ((int)(bid & 63) >= job.gcols * job.grows)
        return;
    if (FMT == HYDK_FMT_F32 && job.rec_bytes != 8) {
        /* the context's token arrays are laid out for 4-byte records; the host widens them before it
         * records a float LF group, so this is unreachable — kept so that a host bug cannot corrupt memory */
        if (threadIdx.x == 0) {
            atomicOr(status, HYDK_STATUS_LAYOUT);
            job.sym_count[bid & 63] = 0;
            job.rbits_total[bid & 63] = 0;
            if (plog)
                part_info[blockIdx.x] = uint2{0u, 0u};
        }
        return;
    }

#if HYDK_K1_CHANSEQ
    static_assert(HYDK_K1_WAVELOCAL, "the channels share one row-pass buffer: a wavefront must read only what it wrote");
    /* [block][kv][kh]: ONE channel's row-pass output at a time, 9.0 KiB; [c][block][kh][kv]: quantised coefficients, 12.0 KiB
     * (integer input: |q| < 2^13, see store_record; float input keeps 32-bit coefficients) */
    typedef typename std::conditional<FMT == HYDK_FMT_F32, int32_t, int16_t>::type coef_t;
    constexpr int kQBlock = 64;
    __shared__ float s_rowpass[kS0Chan];
    __shared__ __attribute__((aligned(16))) coef_t s_q[3 * 32 * kQBlock];
#else
    /* [c][block][kv][kh]: row-pass output, overwritten in place by the quantised coefficients, 27.0 KiB */
    __shared__ float s_rowpass[3 * kS0Chan];
#endif
    /* exclusive prefix sums of the blocks' symbol counts + strip total.  Every wave computes and writes the same
     * 33 values (no barrier needed before it reads them back: its own stores are ordered before its loads) */
    __shared__ uint32_t s_boff[33];
    __shared__ uint32_t s_rbits;                      /* residue bits of the group's symbols */
    /* integer input cannot produce a token above 35 (see store_record): 40 bins per cluster suffice.  LDS is
     * handed out in granules of 1280 bytes (measured: a 33 284-byte build lost the co-residency a 33 232-byte
     * one has): at <= 26 granules two of these workgroups fit beside an entropy-stage workgroup (75 granules) */
    constexpr int kHistW = FMT == HYDK_FMT_F32 ? 72 : (int)HYDK_REC32_TOKENS; /* float input: the (4,1,0) configuration ends at token 71 */
    __shared__ uint32_t s_hist[HYDK_MAX_CLUSTERS * kHistW];
    __shared__ uint16_t s_lut8[256];
    __shared__ uint8_t s_nnz3[64];                    /* coefficient-count context offset (encoder.c:60-66) mod 3 */
    __shared__ uint16_t s_jinfo[64];                  /* zig-zag position j -> natural index kv*8+kh | (frequency context (encoder.c:53-58) mod 3) << 8 */
    /* per (varblock, visit = Y, X, B): {non-zero bitmap by zig-zag position (2 words), symbols | non-zeros << 8,
     * word offset of the block's coefficients in s_rowpass}.  The walk's last step reads one entry past the end
     * and never uses it (whatever follows in LDS, or zero) */
    __shared__ uint4 s_seg[32 * 3];
    __shared__ uint32_t s_blen[32];                   /* symbols of the block's Y | X << 8 | B << 16 runs */
    __shared__ __attribute__((aligned(16))) float s_wq[3 * 64]; /* quantisation weight [channel][kh][kv]: a thread's eight in two 16-byte reads */

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = t >> 6;
    const int g = (int)(bid & 63);
    const int gx = g % job.gcols, gy = g / job.gcols;
    const int px0 = gx << 8, py0 = gy << 8;
    const int gw = min(256, job.width - px0), gh = min(256, job.height - py0);
    const int gbw = (gw + 7) >> 3, gbh = (gh + 7) >> 3;

    for (int i = t; i < HYDK_MAX_CLUSTERS * kHistW; i += kThreads)
        s_hist[i] = 0;
    if (FMT == HYDK_FMT_U8)
        s_lut8[t] = job.in_lut8[t];
    if (t < 64) {
        /* t as a natural index (kv, kh): where it sits in zig-zag order; t as a zig-zag position: its contexts */
        const int j = t;
        s_nnz3[t] = kNnzCtx[t] % 3;
        /* two threads write the two bytes of an entry: kept apart as bytes here, read as one u16 */
        ((uint8_t *)s_jinfo)[2 * kZigzag[t >> 3][t & 7]] = HYDK_K1_CHANSEQ ? (uint8_t)(((t & 7) << 3) | (t >> 3)) : (uint8_t)t; /* (CHANSEQ: coefficients sit [kh][kv]) */
        ((uint8_t *)s_jinfo)[2 * t + 1] = (uint8_t)((j < 2 ? 0 : j < 16 ? j - 1 : j < 32 ? 15 + ((j - 16) >> 1) : 23 + ((j - 32) >> 2)) % 3);
    }
    if (t < 192) /* t = (c, kh, kv) */
        s_wq[t] = (float)kQuantWeight[t >> 6][kZigzag[t & 7][(t >> 3) & 7]];

    /* column / token phases: thread (block cb, horizontal frequency kh) */
    const int cb = t >> 3, kh = t & 7;
    const uint32_t nib_row = (uint32_t)kh * (uint32_t)sizeof(kNibbleMasks.m[0]); /* 256 bytes per kh */
    /* first cluster holding coefficient contexts, by scheme (encoder.c:862-901) */
    const int coef_cl_lo = job.scheme == 0 ? 3 : job.scheme == 3 ? 0 : 1;

    const bool packed = job.pixel_stride == 3 &&
                        (const char *)job.src[1] == (const char *)job.src[0] + sizeof(sample_t) &&
                        (const char *)job.src[2] == (const char *)job.src[0] + 2 * sizeof(sample_t) &&
                        (((uintptr_t)job.src[0] | (uintptr_t)(job.row_stride * (long long)sizeof(sample_t))) & 3) == 0 &&
                        FMT != HYDK_FMT_F32;

    /* phase-A role of this thread: row ar of block ab */
#if HYDK_K1_WAVELOCAL
    /* wavefront w row-transforms all eight rows of blocks 8w .. 8w+7: exactly the blocks whose columns it takes in
     * phase B (cb = t >> 3), so the transposition through LDS stays inside the wavefront */
    const int ar = lane >> 3, ab = (wave << 3) | (lane & 7);
#else
    const int ar = t >> 5, ab = t & 31;
#endif
    const bool fast = packed && ab < gbw && ab * 8 + 8 <= gw; /* whole block row comes as aligned dwords */
    uint32_t nxt[kWords];
#pragma unroll
    for (int k = 0; k < kWords; k++)
        nxt[k] = 0;
    auto prefetch = [&](int s) {
        if (fast && s * 8 + ar < gh) {
            const uint32_t *p = (const uint32_t *)((const char *)job.src[0] +
                                                   ((long long)(py0 + s * 8 + ar) * job.row_stride +
                                                    (long long)(px0 + ab * 8) * 3) * (long long)sizeof(sample_t));
#pragma unroll
            for (int k = 0; k < kWords; k++)
                nxt[k] = HYDK_GLOBAL(const uint32_t, p)[k];
        }
    };
    /* this workgroup's strips [s_first, s_end) and its share of the group's token array */
    const int s_per = (gbh + (1 << plog) - 1) >> plog;
    const int s_first = (int)part * s_per, s_end = min(gbh, s_first + s_per);
    const uint32_t tok_room = job.tok_cap >> plog;
    prefetch(s_first);

    void *const tok = (char *)job.tokens + (size_t)g * job.tok_cap * job.rec_bytes + (size_t)part * tok_room * (FMT == HYDK_FMT_F32 ? 8u : 4u);
    uint32_t goff = 0;
    uint32_t rb_sum = 0;     /* residue bits of the symbols this thread emitted */
    bool overflowed = false; /* the group outgrew its token array: stop storing, report, let the host rerun the frame */
    if (t == 0)
        s_rbits = 0;
    bool bad_sample = false;
    __syncthreads();

    for (int s = s_first; s < s_end; s++) {
        /* ---------------- phase A: 8 px of one block row -> XYB -> row DCT ---------------- */
        float xv[8], yv[8], bv[8]; /* (declared out here: HYDK_K1_CHANSEQ transforms them channel by channel further down) */
        if (ab < gbw) {
            const int y = py0 + s * 8 + ar; /* row inside the LF group */
            const int x0 = px0 + ab * 8;
            const bool row_ok = s * 8 + ar < gh;
            if (fast) {
                uint32_t(&w)[kWords] = nxt; /* this strip's pixels, fetched a strip ago */
                /* which form of the transfer curve this wavefront's 16-bit samples need (wave-uniform) */
                int curve = kCurveBoth;
                if (FMT == HYDK_FMT_U16 && !LUTS) {
                    if (job.linear_light)
                        curve = kCurveNone;
                    else {
                        typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
                        u16x2 mn = __builtin_bit_cast(u16x2, w[0]);
#pragma unroll
                        for (int k = 1; k < kWords; k++)
                            mn = __builtin_elementwise_min(mn, __builtin_bit_cast(u16x2, w[k]));
                        const bool dark = row_ok && (uint32_t)(mn.x < mn.y ? mn.x : mn.y) <= kDarkMax;
                        curve = __builtin_amdgcn_ballot_w64(dark) ? kCurveBoth : kCurveCubic;
                    }
                }
                auto row_to_xyb = [&](auto curve_tag) {
                    constexpr int CURVE = decltype(curve_tag)::value;
#if HYDK_K1_ILP
                    if (FMT != HYDK_FMT_F32 && !LUTS) {
                        constexpr int P = HYDK_K1_ILP; /* pixels evaluated in lock step */
#pragma unroll
                        for (int i0 = 0; i0 < 8; i0 += P) {
                            uint32_t smp[3 * P], lin[3 * P], idx[3 * P];
                            float bia[3 * P];
                            if (FMT == HYDK_FMT_U8) { /* 8-bit samples: the 256-entry transfer table sits in LDS */
#pragma unroll
                                for (int k = 0; k < 3 * P; k++) {
                                    const int si = i0 * 3 + k;
                                    lin[k] = s_lut8[(w[si >> 2] >> (8 * (si & 3))) & 0xFF];
                                }
                            } else {
#pragma unroll
                                for (int k = 0; k < 3 * P; k++) {
                                    const int si = i0 * 3 + k;
                                    smp[k] = (w[si >> 1] >> (16 * (si & 1))) & 0xFFFF;
                                }
                                input_lut16_eval_n<CURVE, 3 * P>(smp, lin);
                            }
#pragma unroll
                            for (int q = 0; q < P; q++) { /* format.c:48-56 */
                                const uint32_t r = lin[3 * q], g = lin[3 * q + 1], b = lin[3 * q + 2];
                                const uint32_t bb = __umul24(5112u, b);
                                idx[3 * q] = umad24(19661u, r, umad24(40761u, g, bb)) >> 16;
                                idx[3 * q + 1] = umad24(15073u, r, umad24(45350u, g, bb)) >> 16;
                                idx[3 * q + 2] = umad24(15953u, r, umad24(13419u, g, __umul24(36163u, b))) >> 16;
                            }
                            /* which of the 3 * P bias values come from the uploaded table through the (otherwise idle)
                             * texture path: HYDK_K1_GATHER bits 3-5 */
                            constexpr auto gathered = [](int k) constexpr { return ((xyb_gmask(XMODE) >> 3) >> (k % 3) & 1) != 0; };
                            constexpr int NG = [&]() constexpr { int n = 0; for (int k = 0; k < 3 * P; k++) n += gathered(k); return n; }();
                            if (NG == 0)
                                bias_lut_eval_n<XMODE, 3 * P>(idx, bia);
                            else {
                                const auto *gl = HYDK_GLOBAL(const float, job.bias_lut);
                                constexpr int NE = 3 * P - NG > 0 ? 3 * P - NG : 1;
                                uint32_t eidx[NE];
                                float eb[NE];
                                int n = 0;
#pragma unroll
                                for (int k = 0; k < 3 * P; k++) {
                                    if (gathered(k))
                                        bia[k] = gl[idx[k]]; /* issued first: they travel while the others are evaluated */
                                    else
                                        eidx[n++] = idx[k];
                                }
                                if (NG < 3 * P)
                                    bias_lut_eval_n<XMODE, NE>(eidx, eb);
                                n = 0;
#pragma unroll
                                for (int k = 0; k < 3 * P; k++)
                                    if (!gathered(k))
                                        bia[k] = eb[n++];
                            }
#pragma unroll
                            for (int q = 0; q < P; q++) {
                                const float Y = (bia[3 * q] + bia[3 * q + 1]) * 0.5f;
                                yv[i0 + q] = Y;
                                xv[i0 + q] = Y - bia[3 * q + 1];
                                bv[i0 + q] = bia[3 * q + 2] - Y;
                            }
                        }
                        return;
                    }
#endif
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        uint32_t rgb[3];
#pragma unroll
                        for (int ch = 0; ch < 3; ch++) {
                            const int si = i * 3 + ch;
                            if (FMT == HYDK_FMT_U8)
                                rgb[ch] = s_lut8[(w[si >> 2] >> (8 * (si & 3))) & 0xFF];
                            else {
                                const uint32_t v = (w[si >> 1] >> (16 * (si & 1))) & 0xFFFF;
                                rgb[ch] = LUTS ? (uint32_t)job.in_lut16[v] : (xyb_gmask(XMODE) >> ch & 1) ? (uint32_t)HYDK_GLOBAL(const uint16_t, job.in_lut16)[v] : input_lut16_eval_as<CURVE>(v);
                            }
                        }
                        if (HYDK_K1_SKIP & 2) { /* timing only: no transfer or bias curves */
                            xv[i] = __uint_as_float(0x3f000000u | w[(i * 3) >> 1]);
                            yv[i] = __uint_as_float(0x3f000000u | w[(i * 3 + 1) >> 1]);
                            bv[i] = __uint_as_float(0x3f000000u | w[(i * 3 + 2) >> 1]);
                            continue;
                        }
                        lms_mix_u16<XMODE>(rgb[0], rgb[1], rgb[2], job.bias_lut, xv[i], yv[i], bv[i]);
                    }
                };
                if (row_ok) {
                    if (curve == kCurveCubic)
                        row_to_xyb(std::integral_constant<int, kCurveCubic>());
                    else if (curve == kCurveNone)
                        row_to_xyb(std::integral_constant<int, kCurveNone>());
                    else
                        row_to_xyb(std::integral_constant<int, kCurveBoth>());
                } else {
#pragma unroll
                    for (int i = 0; i < 8; i++)
                        xv[i] = yv[i] = bv[i] = 0.0f;
                }
                /* the next strip's pixels travel during the transforms and the token phase; issued only now, they
                 * take the registers this strip's pixels have just left */
                prefetch(s + 1);
            } else {
                const int nvalid = row_ok ? min(8, gw - ab * 8) : 0;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    xv[i] = yv[i] = bv[i] = 0.0f; /* edge padding is XYB = 0 (format.c:182-191) */
                    if (i < nvalid) {
                        const long long off = (long long)y * job.row_stride + (long long)(x0 + i) * job.pixel_stride;
                        const sample_t sr = ((const sample_t *)job.src[0])[off];
                        const sample_t sg = ((const sample_t *)job.src[1])[off];
                        const sample_t sb = ((const sample_t *)job.src[2])[off];
                        if (FMT == HYDK_FMT_F32) {
                            if (!lms_mix_f32((float)sr, (float)sg, (float)sb, job.linear_light, xv[i], yv[i], bv[i]))
                                bad_sample = true;
                        } else {
                            uint32_t rgb[3] = {(uint32_t)sr, (uint32_t)sg, (uint32_t)sb};
#pragma unroll
                            for (int ch = 0; ch < 3; ch++) {
                                if (FMT == HYDK_FMT_U8)
                                    rgb[ch] = s_lut8[rgb[ch]];
                                else
                                    rgb[ch] = LUTS ? job.in_lut16[rgb[ch]] : input_lut16_eval(rgb[ch], job.linear_light);
                            }
                            lms_mix_u16<XMODE>(rgb[0], rgb[1], rgb[2], job.bias_lut, xv[i], yv[i], bv[i]);
                        }
                    }
                }
            }
            if (job.dbg_xyb) {
                float *d = job.dbg_xyb + (size_t)y * kDbgPitch + x0;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    d[i] = xv[i];
                    d[(size_t)kDbgPitch * kDbgPitch + i] = yv[i];
                    d[(size_t)2 * kDbgPitch * kDbgPitch + i] = bv[i];
                }
            }
#if !HYDK_K1_CHANSEQ
            float o[8];
            float *dst = s_rowpass + ab * kS0Block + ar * 8;
            dct8(xv, o);
#pragma unroll
            for (int k = 0; k < 8; k++)
                dst[k] = o[k];
            dct8(yv, o);
#pragma unroll
            for (int k = 0; k < 8; k++)
                dst[kS0Chan + k] = o[k];
            dct8(bv, o);
#pragma unroll
            for (int k = 0; k < 8; k++)
                dst[2 * kS0Chan + k] = o[k];
#endif
        }
        /* LDS operations of one wavefront execute in order: its own stores are visible to its loads */
#define HYDK_K1_WAVE_SYNC()                                                                                      \
    do {                                                                                                         \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                                   \
        __builtin_amdgcn_wave_barrier();                                                                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                                   \
    } while (0)
#if HYDK_K1_CHANSEQ
        /* (the row passes come channel by channel, inside phase B) */
#elif HYDK_K1_WAVELOCAL
        HYDK_K1_WAVE_SYNC();
#else
        __syncthreads();
#endif

        /* ---------------- phase B: column DCT, quantise (in place in LDS), LF ints, non-zero bitmaps ---------------- */
        unsigned long long msk[3] = {0, 0, 0}; /* per channel X, Y, B: non-zero coefficients by zig-zag position */
        int32_t lf_int[3] = {0, 0, 0};
        auto column_pass = [&](const int c) __attribute__((always_inline)) {
                float col[8], v[8];
                float *src = s_rowpass + (HYDK_K1_CHANSEQ ? 0 : c * kS0Chan) + cb * kS0Block + kh;
#pragma unroll
                for (int n = 0; n < 8; n++)
                    col[n] = src[n * 8];
                dct8(col, v);
                /* v[kv] = coefficient with vertical frequency kv, horizontal frequency kh; the
                 * reference leaves it at block row kh, column kv (encoder.c:660-664) */
                if (job.dbg_dct) {
                    float *d = job.dbg_dct + (size_t)c * kDbgPitch * kDbgPitch +
                               (size_t)(py0 + s * 8 + kh) * kDbgPitch + px0 + cb * 8;
#pragma unroll
                    for (int kv = 0; kv < 8; kv++)
                        d[kv] = v[kv];
                }
                uint32_t nlo = 0, nhi = 0; /* byte offsets into this kh's nibble-mask rows: bit 3 + b set if coefficient (4 * half + b, kh) is non-zero */
                int q[8];
                const float4 w03 = *(const float4 *)&s_wq[c * 64 + kh * 8], w47 = *(const float4 *)&s_wq[c * 64 + kh * 8 + 4];
                const float wq[8] = {w03.x, w03.y, w03.z, w03.w, w47.x, w47.y, w47.z, w47.w};
#pragma unroll
                for (int kv = 0; kv < 8; kv++) {
                    /* encoder.c:808-811: trunc((coef * weight) * 5); +-1 is the dead zone */
                    const float scaled = v[kv] * wq[kv] * 5.0f;
                    int qq = (int)scaled;
                    /* |trunc(x)| >= 2 exactly when |x| >= 2 (NaN: neither, and converts to 0) */
                    const bool nz = __builtin_fabsf(scaled) >= 2.0f && !(kv == 0 && kh == 0); /* the DC slot is coded by the LF path */
                    qq = nz ? qq : 0;
                    q[kv] = qq;
                    if (kv < 4)
                        nlo |= nz ? 8u << kv : 0u;
                    else
                        nhi |= nz ? 8u << (kv - 4) : 0u;
#if !HYDK_K1_CHANSEQ
                    /* the thread's own column: nobody else reads or writes these eight words */
                    ((int *)src)[kv * 8] = qq;
#endif
                }
#if HYDK_K1_CHANSEQ
                {   /* the thread's eight coefficients [kh][kv = 0..7] in one piece: the wavefront's 64 stores are 1 KiB in a row */
                    coef_t *const dq = s_q + (c * 32 + cb) * kQBlock + kh * 8;
                    if (FMT == HYDK_FMT_F32) {
                        *(int4 *)dq = int4{q[0], q[1], q[2], q[3]};
                        *(int4 *)(dq + 4) = int4{q[4], q[5], q[6], q[7]};
                    } else {
                        const uint32_t k0 = 0x05040100u; /* v_perm_b32: {low half of source 1, low half of source 0} */
                        *(uint4 *)dq = uint4{__builtin_amdgcn_perm((uint32_t)q[1], (uint32_t)q[0], k0), __builtin_amdgcn_perm((uint32_t)q[3], (uint32_t)q[2], k0),
                                             __builtin_amdgcn_perm((uint32_t)q[5], (uint32_t)q[4], k0), __builtin_amdgcn_perm((uint32_t)q[7], (uint32_t)q[6], k0)};
                    }
                }
#endif
                if (job.dbg_quant) {
                    int32_t *d = job.dbg_quant + (size_t)c * kDbgPitch * kDbgPitch +
                                 (size_t)(py0 + s * 8 + kh) * kDbgPitch + px0 + cb * 8;
#pragma unroll
                    for (int kv = 0; kv < 8; kv++)
                        d[kv] = q[kv];
                }
                /* 32-bit offsets from one uniform base: the loads take an SGPR base and a VGPR offset, no 64-bit address arithmetic */
                const char *const nib = (const char *)&kNibbleMasks.m[0][0][0];
                /* the loaded masks are first used after the loop: the next channel's transform runs while they travel */
                if (!(HYDK_K1_SKIP & 4))
                msk[c] = *(const unsigned long long *)(nib + (nib_row | nlo)) | *(const unsigned long long *)(nib + (nib_row | 128u | nhi));
                lf_int[c] = (int32_t)(v[0] * kLfShift[c]); /* LF int: trunc(dc * shift[c]) (encoder.c:573,582) */
        };
#if HYDK_K1_CHANSEQ
        {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                if (ab < gbw) { /* phase A's role: row ar of block ab */
                    float o[8];
                    float *dst = s_rowpass + ab * kS0Block + ar * 8;
                    if (c == 0)
                        dct8(xv, o);
                    else if (c == 1)
                        dct8(yv, o);
                    else
                        dct8(bv, o);
#pragma unroll
                    for (int k = 0; k < 8; k++)
                        dst[k] = o[k];
                }
                HYDK_K1_WAVE_SYNC();
                if (!(HYDK_K1_SKIP & 8) && cb < gbw)
                    column_pass(c);
                HYDK_K1_WAVE_SYNC(); /* the next channel's rows go where this channel's columns were just read */
            }
        }
#endif
        if (!(HYDK_K1_SKIP & 8) && cb < gbw) {
#if !HYDK_K1_CHANSEQ
#pragma unroll
            for (int c = 0; c < 3; c++)
                column_pass(c);
#endif
            if (kh == 0) {
#pragma unroll
                for (int c = 0; c < 3; c++)
                    HYDK_GLOBAL(int32_t, job.dc)[(size_t)c * HYDK_DC_PITCH * HYDK_DC_PITCH + (size_t)((py0 >> 3) + s) * HYDK_DC_PITCH + (px0 >> 3) + cb] = lf_int[c];
            }
#if HYDK_K1_TRIM & 2
            /* the block's bitmap = OR over its eight threads, taken through LDS instead of three DPP steps per word (eighteen
             * vector instructions per thread and strip): the block's eight threads are eight neighbouring lanes of ONE
             * wavefront, whose LDS operations execute in order — thread kh = 0 clears the words, all eight OR into them, thread
             * kh = 0 reads the result back (only it needs it) */
            {
                typedef unsigned long long u64;
                u64 *const cell = (u64 *)&s_seg[cb * 3]; /* .x/.y of the three descriptors: visit order Y, X, B = channel 1, 0, 2 */
                if (kh == 0) {
                    cell[0] = 0;
                    cell[2] = 0;
                    cell[4] = 0;
                }
                __builtin_amdgcn_wave_barrier();
                constexpr int at[3] = {2, 0, 4}; /* channel c -> u64 index (descriptor visit * 2) */
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    __hip_atomic_fetch_or((uint32_t *)&cell[at[c]], (uint32_t)msk[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_or((uint32_t *)&cell[at[c]] + 1, (uint32_t)(msk[c] >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                __builtin_amdgcn_wave_barrier();
                if (kh == 0) {
#pragma unroll
                    for (int c = 0; c < 3; c++)
                        msk[c] = cell[at[c]];
                }
            }
#else
#pragma unroll
            for (int c = 0; c < 3 && !(HYDK_K1_SKIP & 4); c++) { /* the block's bitmap = OR over its eight threads */
                const uint32_t lo = or_reduce8((uint32_t)msk[c]), hi = or_reduce8((uint32_t)(msk[c] >> 32));
                msk[c] = ((unsigned long long)hi << 32) | lo;
            }
#endif
        }
        /* symbols per channel: the count symbol + coefficients up to the last non-zero one; visit order
         * Y, X, B (encoder.c:712) */
        const uint32_t nY = 1u + (msk[1] ? 63u - (uint32_t)__clzll(msk[1]) : 0u);
        const uint32_t nX = 1u + (msk[0] ? 63u - (uint32_t)__clzll(msk[0]) : 0u);
        const uint32_t nB = 1u + (msk[2] ? 63u - (uint32_t)__clzll(msk[2]) : 0u);
        if (kh == 0) {
            s_blen[cb] = cb < gbw ? nY | (nX << 8) | (nB << 16) : 0u;
#if HYDK_K1_CHANSEQ
            const uint32_t base = (uint32_t)(cb * kQBlock), chan_pitch = 32u * kQBlock; /* element offsets into s_q */
#else
            const uint32_t base = (uint32_t)(cb * kS0Block), chan_pitch = (uint32_t)kS0Chan; /* word offsets into s_rowpass */
#endif
            s_seg[cb * 3 + 0] = make_uint4((uint32_t)msk[1], (uint32_t)(msk[1] >> 32), nY | ((uint32_t)__popcll(msk[1]) << 8), base + chan_pitch);
            s_seg[cb * 3 + 1] = make_uint4((uint32_t)msk[0], (uint32_t)(msk[0] >> 32), nX | ((uint32_t)__popcll(msk[0]) << 8), base);
            s_seg[cb * 3 + 2] = make_uint4((uint32_t)msk[2], (uint32_t)(msk[2] >> 32), nB | ((uint32_t)__popcll(msk[2]) << 8), base + 2 * chan_pitch);
        }
        __syncthreads();

        /* ---------------- phase C1: every wave prefix-sums the 32 block totals for itself ---------------- */
        {
            const uint32_t len3 = s_blen[lane & 31];
            const uint32_t mine = (len3 & 0xffu) + ((len3 >> 8) & 0xffu) + (len3 >> 16);
            uint32_t inc = scan16_inclusive(mine);
            inc += HYDK_DPP(inc, 0x142, 0xA); /* row_bcast:15: rows 1 and 3 add the total of the row before */
            s_boff[(lane & 31) + 1] = inc;
            if (lane == 0)
                s_boff[0] = 0;
            __builtin_amdgcn_wave_barrier();
        }
        const uint32_t strip_total = s_boff[32];

        /* ---------------- phase C2: the strip's symbol stream, cut into 256 equal runs ----------------
         * Thread t emits one of 256 consecutive runs of the strip's symbols, equal in length up to one: it finds the
         * (block, channel, zig-zag position) its run starts at by a binary search over the block offsets, then WALKS
         * — the non-zero count still to come and the "previous coefficient was non-zero" flag of the contexts
         * (encoder.c:724-738) are carried from symbol to symbol instead of being recounted from the bitmap, and the
         * work is balanced over the workgroup whatever the blocks' sizes. */
        {
            overflowed = overflowed || goff + strip_total > tok_room;
            /* runs of floor(n / 256) symbols, the first n mod 256 threads one more: the longer runs sit in the first
             * wavefronts, so the later ones leave the loop an iteration earlier */
            static_assert(kThreads == 256, "run lengths are computed with shifts");
            const uint32_t per = strip_total >> 8, extra = strip_total & 255u;
            uint32_t p = (uint32_t)t * per + min((uint32_t)t, extra);
            const uint32_t pend = overflowed ? 0u : p + per + ((uint32_t)t < extra ? 1u : 0u);
            if (!(HYDK_K1_SKIP & 1) && p < pend) {
                uint32_t b = 0;
#pragma unroll
                for (uint32_t step = 16; step; step >>= 1)
                    b += s_boff[b + step] <= p ? step : 0u;
                uint32_t j = p - s_boff[b]; /* position in the block's symbols, then in the channel's */
                const uint32_t len = s_blen[b];
                uint32_t visit = 0;
                if (j >= (len & 0xffu)) {
                    j -= len & 0xffu;
                    visit = 1;
                    if (j >= ((len >> 8) & 0xffu)) {
                        j -= (len >> 8) & 0xffu;
                        visit = 2;
                    }
                }
                uint32_t seg_at = b * 3u + visit;
                uint4 seg = s_seg[seg_at];
                uint32_t n = seg.z & 0xffu, nz_total = seg.z >> 8;
                /* non-zeros at positions >= j; whether position j - 1 holds one (encoder.c:731-738) */
                const unsigned long long m = ((unsigned long long)seg.y << 32) | seg.x;
                uint32_t remaining = nz_total - (uint32_t)__popcll(m & ((1ull << j) - 1ull));
                uint32_t prev = j <= 1u ? (uint32_t)(nz_total <= 4u) : (uint32_t)(m >> (j - 1u)) & 1u;
                /* cluster of a coefficient = coef_base + (prev & prev_mask) + (2 * ((visit + nnz ctx + freq ctx) mod 3) & ctx_mask),
                 * of a count = visit & count_mask: the four schemes of encoder.c:862-901 without a branch */
                const uint32_t coef_base = (uint32_t)coef_cl_lo;
                const uint32_t prev_mask = job.scheme <= 1 ? 1u : 0u, ctx_mask = job.scheme == 0 ? 6u : 0u;
                const uint32_t count_mask = job.scheme == 0 ? 3u : 0u;
                for (; p < pend; p++) {
                    const uint32_t ji = s_jinfo[j];
#if HYDK_K1_CHANSEQ
                    const int coef = (int)s_q[seg.w + (ji & 0xffu)];
#else
                    const int coef = ((const int *)s_rowpass)[seg.w + (ji & 0xffu)];
#endif
                    const bool is_count = j == 0u;
                    const uint32_t value = is_count ? nz_total : pack_signed(coef);
                    /* context - 111 = 458*visit + prev + 2*(nnz_ctx[remaining] + freq_ctx[j]) (encoder.c:724,731-732);
                     * scheme 0 needs it mod 6 = prev + 2*((visit + nnz_ctx + freq_ctx) mod 3), the others mod 2 = prev.
                     * u = 0..6; 2 * (u mod 3) sits at bits 2u+1, 2u+2 of 0x1248 */
                    const uint32_t u = visit + (uint32_t)s_nnz3[remaining & 63u] + (ji >> 8);
                    const uint32_t cluster = is_count ? (visit & count_mask)
                                                      : coef_base + (prev & prev_mask) + ((0x1248u >> (u + u)) & ctx_mask);
                    /* hybrid-uint split, config (4,1,0) (entropy.c:427-444) */
                    uint32_t token, rbits, residue;
                    if (FMT != HYDK_FMT_F32) {
                        /* value < 2^14 converts exactly: its float's exponent and top mantissa bit are floor(log2) and
                         * the bit below the leading one, i.e. the token's two variable parts (entropy.c:427-444) */
                        const uint32_t fb = __float_as_uint((float)value) >> 22; /* 2 * (127 + floor(log2)) + next bit */
                        const bool big = value >= 16u;
                        token = big ? fb - 246u : value;
                        rbits = big ? (fb >> 1) - 128u : 0u;
                        residue = __builtin_amdgcn_ubfe(value, 0u, rbits);
                    } else if (value < 16) {
                        token = value;
                        rbits = 0;
                        residue = 0;
                    } else {
                        const int nb = 30 - __clz((int)value); /* floor(log2) - 1 */
                        rbits = (uint32_t)nb;
                        residue = value & ((1u << nb) - 1u);
                        token = 16u + (((uint32_t)(nb - 3) << 1) | ((value >> nb) & 1u));
                    }
#if HYDK_K1_TRIM
                    const uint32_t bin = umad24(cluster, (uint32_t)kHistW, FMT == HYDK_FMT_F32 ? min(token, (uint32_t)kHistW - 1u) : token);
#else
                    const uint32_t bin = cluster * kHistW + (FMT == HYDK_FMT_F32 ? min(token, (uint32_t)kHistW - 1u) : token);
#endif
                    store_record<FMT>(tok, goff + p, token, cluster, bin, rbits, residue);
                    rb_sum += rbits;
                    /* every token straight into the LDS histogram (until round 3 zero tokens were counted in packed per-thread
                     * counters inside a divergent branch: twelve instructions to save an atomic that costs nothing) */
                    atomicAdd(&s_hist[bin], 1u);
                    /* walk on: the DC slot read for a count symbol is zero, so it leaves `remaining` alone */
                    const uint32_t here = coef != 0 ? 1u : 0u;
                    remaining -= here;
                    prev = is_count ? (uint32_t)(nz_total <= 4u) : here;
                    if (++j == n) { /* next channel of the block, or the next block (at most one step: every run has its count symbol) */
                        seg_at++;
                        visit = visit == 2u ? 0u : visit + 1u;
                        seg = s_seg[seg_at];
                        n = seg.z & 0xffu;
                        nz_total = remaining = seg.z >> 8;
                        j = 0;
                    }
                }
            }
        }
        goff += strip_total;
        /* the next strip's row pass overwrites the coefficients this strip's token phase reads */
        __syncthreads();
    }

    __syncthreads();
    uint32_t top_token = 0;
    for (int i = t; i < HYDK_MAX_CLUSTERS * kHistW; i += kThreads) {
        const uint32_t v = s_hist[i];
        if (v) {
            atomicAdd(&job.hist[(i / kHistW) * HYDK_ALPHABET + i % kHistW], v);
            top_token = max(top_token, (uint32_t)(i % kHistW) + 1u);
        }
    }
#pragma unroll
    for (int d = 32; d; d >>= 1)
        top_token = max(top_token, (uint32_t)__shfl_xor((int)top_token, d));
    if (lane == 0 && top_token)
        atomicMax(job.alpha_max, top_token);
#pragma unroll
    for (int d = 32; d; d >>= 1)
        rb_sum += (uint32_t)__shfl_xor((int)rb_sum, d);
    if (lane == 0)
        atomicAdd(&s_rbits, rb_sum);
    __syncthreads();
    if (t == 0) {
        const uint32_t count = overflowed || HYDK_K1_SKIP ? 0u : goff; /* timing-only builds leave no symbols for the later stages */
        if (plog) {
            part_info[blockIdx.x] = uint2{count, s_rbits};
        } else {
            job.sym_count[g] = count;
            job.rbits_total[g] = s_rbits;
        }
        if (overflowed)
            atomicOr(status, HYDK_STATUS_TOKENS);
    }
    if (FMT == HYDK_FMT_F32 && bad_sample)
        atomicOr(status, HYDK_STATUS_BAD_SAMPLE);
}

#include "lf_huffman.h" /* the LF coder's code construction rides in the table kernel's (or the chain kernel's) launch, see below */

/* a group coded by several transform workgroups (k_transform_tokenize, plog > 0): part q's records sit at q * (tok_cap >> plog)
 * of the group's token array; moved down so that the array is the group's symbol stream in one piece, and the group's symbol
 * and residue-bit totals written where the single-workgroup form leaves them.  One workgroup per group; a part moves towards
 * lower addresses only, chunk by chunk (a chunk is read by all threads before any of it is written, and never reaches into
 * the chunk after it).  grid = 64 x LF groups of the launch. */
__global__ __launch_bounds__(kThreads) void k_join_parts(const HydkLfJob *__restrict__ jobs, const uint2 *part_info, int plog,
                                                         uint32_t *status) {
    const HydkLfJob job = jobs[blockIdx.x >> 6];
    const int g = blockIdx.x & 63;
    if (g >= job.gcols * job.grows)
        return;
    /* records are 8 bytes for float LF groups and 4 for integer ones — also in a context whose arrays were laid out for
     * 8-byte records by an earlier float frame (job.rec_bytes is the group PITCH's unit, not this group's record size) */
    const uint32_t parts = 1u << plog, room = job.tok_cap >> plog, wpr = job.fmt == HYDK_FMT_F32 ? 2u : 1u; /* 32-bit words per record */
    uint32_t *const base = (uint32_t *)((char *)job.tokens + (size_t)g * job.tok_cap * job.rec_bytes);
    /* a part's count is a COPY LENGTH here: one that no transform workgroup wrote (the slot's descriptor named a kernel
     * variant the launch did not contain — a host bug) or that exceeds the part's share must not move memory */
    bool bogus = false;
    uint2 p0 = part_info[(size_t)blockIdx.x * parts];
    if (p0.x > room) {
        p0 = uint2{0u, 0u};
        bogus = true;
    }
    uint32_t total = p0.x, rbits = p0.y;
    for (uint32_t q = 1; q < parts; q++) {
        uint2 pi = part_info[(size_t)blockIdx.x * parts + q];
        if (pi.x > room) {
            pi = uint2{0u, 0u};
            bogus = true;
        }
        const uint32_t *src = base + (size_t)q * room * wpr;
        uint32_t *dst = base + (size_t)total * wpr;
        const uint32_t words = pi.x * wpr;
        if (dst != src) {
            constexpr uint32_t kPer = 8; /* words a thread carries per chunk: eight loads in flight, one barrier per 2048 words */
            for (uint32_t at = 0; at < words; at += kPer * kThreads) { /* (uniform trip count: the barrier is reached by all) */
                uint32_t v[kPer];
#pragma unroll
                for (uint32_t k = 0; k < kPer; k++) {
                    const uint32_t i = at + k * kThreads + threadIdx.x;
                    v[k] = i < words ? src[i] : 0u;
                }
                __syncthreads();
#pragma unroll
                for (uint32_t k = 0; k < kPer; k++) {
                    const uint32_t i = at + k * kThreads + threadIdx.x;
                    if (i < words)
                        dst[i] = v[k];
                }
            }
        }
        total += pi.x;
        rbits += pi.y;
    }
    if (threadIdx.x == 0) {
        job.sym_count[g] = bogus ? 0u : total;
        job.rbits_total[g] = bogus ? 0u : rbits;
        if (bogus)
            atomicOr(status, HYDK_STATUS_INCONSISTENT);
    }
}

/* ==========================================================================================
 * K2: per-LF-group ANS tables.  grid = LF groups of the frame (send order), block = 256.
 * With lf_hist != NULL the launch carries num_slots PASSENGER workgroups: workgroup num_slots + s builds the prefix code of
 * LF group s's coefficient stream (lf_huffman.h: 200 us of one wavefront, which depends only on the LF token histograms
 * and not on this kernel's tables).  An A/B switch (HYDAMD_LF_CODES_RIDE=tables): by default the passengers ride in the
 * chain kernel's launch, which lasts 2.5 ms anyway.
 * ======================================================================================== */
__global__ __launch_bounds__(kThreads) void k_build_tables(const uint32_t *hist_all, HydkTables *tabs,
                                                           const uint32_t *alpha_max_all, int nclusters,
                                                           uint32_t alpha_floor, const uint32_t *alpha_floor_dev,
                                                           int first_slot, int num_slots, const uint32_t *lf_hist,
                                                           HydkLfStream *lf_streams, void *lf_work, int slots_per_frame) {
    HYDK_URGENT();
    if ((int)blockIdx.x >= num_slots) {
        __shared__ LfHuffScratch s_huff;
        const int s = (int)blockIdx.x - num_slots;
        if (threadIdx.x < 64)
            lf_huffman_wave(lf_hist + (size_t)s * HYDK_LF_CODES, ((LfWork *)lf_work)[s].codes, lf_streams + s, s_huff, (int)threadIdx.x);
        return;
    }
    const unsigned slot = (unsigned)first_slot + blockIdx.x; /* all arrays are indexed by the frame's slot */
    const uint32_t *hist = hist_all + (size_t)slot * HYDK_MAX_CLUSTERS * HYDK_ALPHABET;
    HydkTables *tab = tabs + slot;
    __shared__ uint32_t s_freq[HYDK_MAX_CLUSTERS][HYDK_ALPHABET];
    __shared__ uint32_t s_cutoff[HYDK_MAX_CLUSTERS][HYDK_ALPHABET];
    __shared__ uint32_t s_other[HYDK_MAX_CLUSTERS][HYDK_ALPHABET];
    __shared__ uint32_t s_shift[HYDK_MAX_CLUSTERS][HYDK_ALPHABET];
    __shared__ uint32_t s_base[HYDK_MAX_CLUSTERS][HYDK_ALPHABET];
    __shared__ uint8_t s_under[HYDK_MAX_CLUSTERS][HYDK_ALPHABET];
    __shared__ uint8_t s_over[HYDK_MAX_CLUSTERS][HYDK_ALPHABET];
    __shared__ uint32_t s_alpha[HYDK_MAX_CLUSTERS];
    __shared__ uint32_t s_log_alpha, s_err;

    const int t = threadIdx.x;
    for (int i = t; i < HYDK_MAX_CLUSTERS * HYDK_ALPHABET; i += kThreads) {
        (&s_freq[0][0])[i] = hist[i];
        (&s_cutoff[0][0])[i] = 0;
        (&s_other[0][0])[i] = 0;
        (&s_shift[0][0])[i] = 0;
        (&s_base[0][0])[i] = 0;
    }
    if (t == 0)
        s_err = 0;
    __syncthreads();

    if (t < HYDK_MAX_CLUSTERS) {
        uint32_t a = 0;
        if (t < nclusters)
            for (int k = 0; k < HYDK_ALPHABET; k++)
                if (s_freq[t][k])
                    a = k + 1;
        s_alpha[t] = a;
    }
    __syncthreads();
    if (t == 0) {
        /* the stream-wide alphabet maximum is never reset between LF groups (entropy.c:459-460,952):
         * LF group n codes with the maximum over LF groups 0..n in send order */
        uint32_t mx = alpha_floor; /* maximum over the LF groups other GPUs coded before ours */
        if (alpha_floor_dev)       /* ... when it was exchanged on the device (no host round trip) */
            mx = max(mx, *alpha_floor_dev);
        /* a context that codes a BATCH of independent frames in one launch group (hydamd_begin_batch) holds frame k in slots
         * k * slots_per_frame ...: the maximum starts afresh with every frame */
        const unsigned frame_first = slots_per_frame > 0 ? slot - slot % (unsigned)slots_per_frame : 0u;
        for (unsigned sl = frame_first; sl <= slot; sl++)
            mx = max(mx, alpha_max_all[sl]);
        uint32_t lg = mx > 1 ? 32 - __clz((int)(mx - 1)) : 0; /* ceil(log2(mx)) */
        s_log_alpha = max(lg, 5u);
        tab->running_max_alphabet = mx;
    }
    __syncthreads();
    const int log_alpha = (int)s_log_alpha;
    const int log_bucket = 12 - log_alpha;
    const uint32_t bucket = 1u << log_bucket, table_size = 1u << log_alpha;

    if (t < nclusters && s_alpha[t]) {
        uint32_t *f = s_freq[t];
        const uint32_t n = s_alpha[t];
        /* ---- 12-bit normalisation (entropy.c:267-301) ---- */
        unsigned long long total = 0;
        for (uint32_t k = 0; k < n; k++)
            total += f[k];
        unsigned long long scaled = 0;
        for (uint32_t k = 0; k < n; k++) {
            if (!f[k])
                continue;
            uint32_t v = (uint32_t)((((unsigned long long)f[k] << 12) / total) & 0xFFFFu);
            if (!v)
                v = 1;
            f[k] = v;
            scaled += v;
        }
        int j = (int)n - 1;
        while (scaled > 4096 && j >= 0) {
            const unsigned long long excess = scaled - 4096;
            if (excess < f[j]) {
                f[j] -= (uint32_t)excess;
                scaled -= excess;
                break;
            } else if (f[j] > 1) {
                scaled -= f[j] - 1;
                f[j] = 1;
            }
            j--;
        }
        f[0] += (uint32_t)(4096 - scaled);
        const bool unique = f[n - 1] == 4096;

        /* ---- alias table (entropy.c:184-242) ---- */
        uint32_t *cutoff = s_cutoff[t], *other = s_other[t], *shift = s_shift[t];
        if (unique) {
            for (uint32_t i = 0; i < table_size; i++) {
                other[i] = n - 1;
                shift[i] = i * bucket;
            }
        } else {
            uint8_t *under = s_under[t], *over = s_over[t];
            uint32_t nu = 0, no = 0;
            for (uint32_t sidx = 0; sidx < n; sidx++) {
                cutoff[sidx] = f[sidx];
                if (f[sidx] < bucket)
                    under[nu++] = (uint8_t)sidx;
                else if (f[sidx] > bucket)
                    over[no++] = (uint8_t)sidx;
            }
            for (uint32_t sidx = n; sidx < table_size; sidx++)
                under[nu++] = (uint8_t)sidx;
            while (no) {
                if (!nu) {
                    s_err = 1;
                    break;
                }
                const uint32_t u = under[--nu], o = over[--no];
                const uint32_t moved = bucket - cutoff[u];
                cutoff[o] -= moved;
                shift[u] = cutoff[o];
                other[u] = o;
                if (cutoff[o] < bucket)
                    under[nu++] = (uint8_t)o;
                else if (cutoff[o] > bucket)
                    over[no++] = (uint8_t)o;
            }
            for (uint32_t i = 0; i < table_size; i++) {
                if (cutoff[i] == bucket) {
                    other[i] = i;
                    cutoff[i] = shift[i] = 0;
                } else {
                    shift[i] -= cutoff[i];
                }
            }
        }
        uint32_t run = 0;
        for (uint32_t k = 0; k < HYDK_ALPHABET; k++) {
            s_base[t][k] = run;
            run += k < n ? f[k] : 0;
        }
    }
    __syncthreads();

    /* ---- inverse slot table: evaluate the decoder-direction alias map for every slot; it is a
     * bijection onto (symbol, offset < freq), which is what the reference's per-symbol search
     * (entropy.c:1104-1113) inverts ---- */
    for (int idx = t; idx < nclusters * HYDK_ANS_SLOTS; idx += kThreads) {
        const int c = idx >> 12;
        const uint32_t slot = idx & (HYDK_ANS_SLOTS - 1);
        if (!s_alpha[c])
            continue;
        const uint32_t i = slot >> log_bucket, pos = slot & (bucket - 1);
        uint32_t sym, off;
        if (pos >= s_cutoff[c][i]) {
            sym = s_other[c][i];
            off = s_shift[c][i] + pos;
        } else {
            sym = i;
            off = pos;
        }
        if (sym >= HYDK_ALPHABET || off >= s_freq[c][sym]) {
            s_err = 2;
            continue;
        }
        const uint32_t at = 2u * s_base[c][sym] + off, f = s_freq[c][sym];
        tab->inv[c][at] = (uint16_t)slot;
        tab->inv[c][at + f] = (uint16_t)(slot + HYDK_ANS_SLOTS);
        tab->inv1[c][s_base[c][sym] + off] = (uint16_t)slot;
    }
    for (int idx = t; idx < HYDK_MAX_CLUSTERS * HYDK_ALPHABET; idx += kThreads) {
        const int c = idx / HYDK_ALPHABET, k = idx % HYDK_ALPHABET;
        const bool live = c < nclusters && s_alpha[c] != 0;
        const uint32_t f = live ? s_freq[c][k] : 0;
        tab->freq[c][k] = f;
        tab->fb[c][k] = f | (s_base[c][k] << 16);
        tab->magic[c][k] = f <= 1 ? 0xFFFFFFFFu : (uint32_t)(0x100000000ull / f);
    }
    __syncthreads();
    if (t < HYDK_MAX_CLUSTERS)
        tab->alphabet[t] = s_alpha[t];
    if (t == 0) {
        tab->log_alphabet_size = s_log_alpha;
        tab->error = s_err;
    }
}

/* ==========================================================================================
 * K3a: reverse rANS.  One wave per group, 4 groups of one LF group per workgroup (they share the
 * preset's tables in LDS).  The state -> state recurrence is strictly serial per group
 * (entropy.c:1092-1119), so the kernel is shaped around it:
 *   - 64 symbols at a time are loaded coalesced and their (freq, magic, table base) looked up by
 *     all lanes in parallel;
 *   - the serial walk then touches only the recurrence: every lane carries the same state
 *     redundantly (no divergence, LDS reads broadcast), per-symbol operands arrive by
 *     v_readlane, and lane k keeps the state seen by symbol k;
 *   - refill decisions and all bit emission are recomputed wave-parallel from those saved states:
 *     prefix-sum of bit counts, OR into an LDS window, coalesced store.  The group buffer is
 *     filled from its END so the forward order [state][refill_p][residue_p].. of
 *     entropy.c:1127-1147 falls out of the reverse walk without a replay pass.
 * grid = 16 x LF groups, block = 256.
 * ======================================================================================== */
constexpr int kWinWords = 100; /* 64 symbols x 46 bits = 92 words + alignment slack */

/* record p of a group in the 8-byte form {lo = token | cluster << 8 | bit count << 16, hi = residue},
 * whichever form K1 wrote (hydk_common.h); p < 0 reads nothing */
__device__ __forceinline__ uint2 load_record(const void *tok, int p, bool wide) {
    uint2 r = {0u, 0u};
    if (p >= 0) {
        /* device memory, said so: a flat load counts against the LDS counter too, and every slot lookup of the walk
         * would wait for the records that are meant to travel during it */
        if (wide) {
            typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
            const u32x2 v = HYDK_GLOBAL(const u32x2, tok)[p];
            r = uint2{v.x, v.y};
        } else {
            const uint32_t v = HYDK_GLOBAL(const uint32_t, tok)[p];
            r.x = HYDK_REC32_TO_LO(v);
            r.y = v >> 16;
        }
    }
    return r;
}
/* four 4-byte records, 16-byte aligned, as a GLOBAL load (the token arrays' address comes out of the job descriptor,
 * generic to the compiler: a flat load, which every LDS wait of a walk would then wait for as well) */
typedef uint32_t hydk_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 load_records4(const void *tok, int i) {
    const hydk_u32x4 v = HYDK_GLOBAL(const hydk_u32x4, tok)[i];
    return uint4{v.x, v.y, v.z, v.w};
}
constexpr int kInvEntries = HYDK_MAX_CLUSTERS * 2 * HYDK_ANS_SLOTS;

/* x = state > thr ? state >> 16 : state in two issue slots: the select reads the high half of the
 * state through SDWA instead of a separate shift (every slot counts: the chain is one wave issuing
 * in order, so an instruction off the dependency path still delays the next step) */
__device__ __forceinline__ uint32_t rans_renorm(uint32_t state, uint32_t thr) {
    uint32_t x;
    asm("v_cmp_gt_u32 vcc, %1, %2\n\t"
        "v_cndmask_b32_sdwa %0, %1, %1, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"
        : "=v"(x)
        : "v"(state), "v"(thr)
        : "vcc");
    return x;
}
/* a + b * c with 24-bit factors in one slot (the compiler prefers a multiply and a three-input add) */
__device__ __forceinline__ uint32_t mad24(uint32_t b, uint32_t c, uint32_t a) {
    uint32_t r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(c), "v"(a));
    return r;
}

__device__ __forceinline__ uint64_t mad_u64_u32(uint32_t a, uint32_t b, uint64_t c) {
    uint64_t d;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c) : "vcc");
    return d;
}
__device__ __forceinline__ uint64_t mul_u64_u32(uint32_t a, uint32_t b) {
    uint64_t d;
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b) : "vcc");
    return d;
}
__device__ __forceinline__ uint32_t mad_u24(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

/* one step of the recurrence; the symbol's operands {-2f, floor(2^32/f), table address, threshold}
 * were staged in LDS by the lane that owns it and arrive by one broadcast ds_read_b128 */
typedef __attribute__((address_space(3))) const uint16_t HydkLdsU16;
#define HYDK_RANS_STEP(o)                                                                     \
    do {                                                                                      \
        /* lane 0 <- state, lane l <- trail[l-1]: the states file past, newest in lane 0 */   \
        trail = (uint32_t)__builtin_amdgcn_update_dpp((int)state, (int)trail, 0x138, 0xF, 0xF, false); \
        const uint32_t x = rans_renorm(state, o.w);                                           \
        /* q = mulhi(x, floor(2^32/f)) is floor(x/f) or one less, so r = x - q*f < 2f; the   \
         * doubled table returns slot(r mod f) + 4096*(r >= f), which also repairs q.  The    \
         * LDS address adr + 2r is formed as (adr + 2x) + q*(-2f): one op after the mulhi */  \
        const uint32_t q = __umulhi(x, o.y);                                                  \
        const uint32_t at = mad24(q, o.x, o.z + 2u * x);                                      \
        const uint32_t ent = *(HydkLdsU16 *)(uintptr_t)at;                                    \
        state = (q << 12) + ent;                                                              \
    } while (0)

template <int WAVES, bool DEFER> /* groups (= waves) per workgroup; DEFER: the bits are written by k_rans_emit */
__global__ __launch_bounds__(64 * WAVES) void k_rans_encode(const HydkLfJob *__restrict__ jobs, const uint32_t *sym_count_all,
                                                            const HydkTables *tabs, uint32_t *bitbuf_all,
                                                            uint32_t bit_pitch_words, uint32_t *group_bits_all,
                                                            int preset_bits, const uint32_t *status, uint16_t *aux_all,
                                                            uint16_t *flags_all, uint32_t aux_pitch, uint32_t *final_state_all) {
    constexpr int kThreads = 64 * WAVES;                             /* shadows the file-level constant */
    constexpr int kBlocksPerLfg = HYDK_GROUPS_PER_LFG / WAVES;
    __shared__ uint16_t s_inv[kInvEntries];                          /* 144 KiB */
    __shared__ uint32_t s_fb[HYDK_MAX_CLUSTERS * HYDK_ALPHABET];
    __shared__ uint32_t s_magic[HYDK_MAX_CLUSTERS * HYDK_ALPHABET];
    __shared__ uint32_t s_win[WAVES][DEFER ? 1 : kWinWords];           /* the bit writer's window (not with DEFER) */
    __shared__ uint4 s_ops[WAVES][64];                               /* per-symbol operands of the chunk being walked */

    HYDK_URGENT();
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int slot = blockIdx.x / kBlocksPerLfg;
    const int first_group = (blockIdx.x % kBlocksPerLfg) * WAVES;
    const int g = first_group + wave;
    const int ngroups = jobs[slot].gcols * jobs[slot].grows;
    if (*status & HYDK_STATUS_OVERFLOW)
        return; /* the transform stage ran out of token space: the host reruns the frame */
    if (first_group >= ngroups) {
        if (lane == 0)
            group_bits_all[slot * HYDK_GROUPS_PER_LFG + g] = 0;
        return; /* whole workgroup beyond the LF group's last group */
    }
    const HydkTables *tab = tabs + slot;
    {
        const uint4 *src = (const uint4 *)&tab->inv[0][0];
        uint4 *dst = (uint4 *)s_inv;
        for (int i = t; i < kInvEntries / 8; i += kThreads)
            dst[i] = src[i];
        for (int i = t; i < HYDK_MAX_CLUSTERS * HYDK_ALPHABET; i += kThreads) {
            s_fb[i] = (&tab->fb[0][0])[i];
            s_magic[i] = (&tab->magic[0][0])[i];
        }
    }
    __syncthreads();
    if (g >= ngroups) {
        if (lane == 0)
            group_bits_all[slot * HYDK_GROUPS_PER_LFG + g] = 0;
        return;
    }

    const size_t G = (size_t)slot * HYDK_GROUPS_PER_LFG + g;
    const bool wide = jobs[slot].fmt == HYDK_FMT_F32;
    const void *tok = (const char *)jobs[slot].tokens + (size_t)g * jobs[slot].tok_cap * jobs[slot].rec_bytes;
    uint32_t *W = bitbuf_all + G * bit_pitch_words;
    uint32_t *win = s_win[wave];
    uint4 *ops = s_ops[wave];
    const int n = __builtin_amdgcn_readfirstlane((int)sym_count_all[G]); /* wave-uniform: scalar loop control */
    const uint32_t inv_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t *)s_inv;

    uint32_t cur = bit_pitch_words * 32u; /* stream start so far (absolute bit position) */
    uint32_t carry = 0;                   /* content of the partly filled word at cur>>5 */
    uint32_t state;
    /* keep the recurrence in vector registers: routed through the scalar unit, every LDS lookup
     * would cost a v_mov + v_readfirstlane round trip on the critical path */
    asm volatile("v_mov_b32 %0, 0x130000" : "=v"(state));

    /* wave-parallel emission of one (value, nbits) per lane, lane 0 nearest the already written
     * bits (= latest in stream order), lane 63 earliest */
    auto emit = [&](unsigned long long val, uint32_t nbits) {
        const uint32_t inc = scan64_inclusive(nbits);
        const uint32_t total = __builtin_amdgcn_readlane(inc, 63);
        if (!total)
            return;
        const uint32_t newcur = cur - total;
        const uint32_t wlo = newcur >> 5, whi = (cur - 1u) >> 5;
        const uint32_t nwords = whi - wlo + 1u;
        for (uint32_t i = lane; i < nwords; i += 64)
            win[i] = 0;
        __builtin_amdgcn_wave_barrier();
        if (lane == 0 && (cur & 31u))
            win[whi - wlo] = carry;
        /* win[] is private to this wave: lock-step execution + in-order LDS make the phases
         * visible to each other; the wave barriers only pin the compiler's ordering */
        __builtin_amdgcn_wave_barrier();
        if (nbits) {
            const uint32_t pos = cur - inc - wlo * 32u; /* cur - (exclusive prefix) - nbits, window-relative */
            const uint32_t w = pos >> 5, sh = pos & 31u;
            const unsigned long long lo = val << sh;
            const uint32_t hi = sh ? (uint32_t)(val >> (64u - sh)) : 0u;
            if ((uint32_t)lo)
                atomicOr(&win[w], (uint32_t)lo);
            if ((uint32_t)(lo >> 32))
                atomicOr(&win[w + 1], (uint32_t)(lo >> 32));
            if (hi)
                atomicOr(&win[w + 2], hi);
        }
        __builtin_amdgcn_wave_barrier();
        const bool low_partial = (newcur & 31u) != 0;
        for (uint32_t i = lane + (low_partial ? 1u : 0u); i < nwords; i += 64)
            W[wlo + i] = win[i];
        carry = low_partial ? win[0] : 0u;
        __builtin_amdgcn_wave_barrier();
        cur = newcur;
    };

    /* chunks of 64 symbols aligned to the START of the token array: the chunk that holds the stream's end comes first
     * and is the partial one (so that the refill flags of a chunk are whole 16-bit words of the flag array) */
    uint16_t *aux = DEFER ? aux_all + G * aux_pitch : nullptr;
    uint16_t *flags = DEFER ? flags_all + G * (aux_pitch / 16) : nullptr;
    uint32_t refills = 0;
    uint2 rec_next = load_record(tok, lane <= ((n - 1) & 63) ? n - 1 - lane : -1, wide);
    for (int hi_p = n - 1; hi_p >= 0; hi_p = (hi_p & ~63) - 1) {
        /* lane l owns symbol p = hi_p - l; the walk visits lanes 0, 1, 2, ... */
        const int cnt = (hi_p & 63) + 1;
        const int p = hi_p - lane;
        const bool valid = lane < cnt;
        const uint2 rec = rec_next;
        rec_next = load_record(tok, (hi_p & ~63) - 1 - lane, wide); /* the next chunk's records (a full chunk, or none) travel during this chunk's walk */
        const uint32_t lo = rec.x;
        const uint32_t e = ((lo >> 8) & 0xF) * HYDK_ALPHABET + (lo & 0xFF);
        const uint32_t fbv = s_fb[e];
        const uint32_t f = valid ? (fbv & 0xFFFFu) : 1u;
        /* (state >> 20) >= f  <=>  state > (f << 20) - 1, exact for f up to 4096 (entropy.c:1092) */
        const uint32_t thr = (uint32_t)(((unsigned long long)f << 20) - 1ull);
        uint4 op;
        op.x = (uint32_t)(-2 * (int)f);
        op.y = s_magic[e];
        op.z = inv_lds + (((lo >> 8) & 0xF) * (2u * HYDK_ANS_SLOTS) + 2u * (fbv >> 16)) * 2u; /* the LDS address of the symbol's doubled slot list */
        op.w = thr;
        ops[lane] = op;
        __builtin_amdgcn_wave_barrier();
        uint32_t trail = 0;
        if (cnt == 64) {
            /* eight steps a block; the next block's operands are requested before this block's walk, so that no
             * walk starts with an LDS round trip and no slot lookup waits behind operand reads.
             * Round 5: what bounds a step is BOTH its dependent depth (~9.5 cycles per dependent instruction plus the lookup's
             * ~64) and its instruction count (~6.7 cycles each for a wavefront alone on its SIMD) — cutting the depth to two
             * by forming 64-bit partial products ahead cost 19 instructions and 30 % (profiles/r05_chain_anatomy.txt).  So only
             * the cheap part moves behind the lookup's request: the refill test of the NEXT symbol (on bits 12.. of the new
             * state: it needs the repaired quotient, not the slot) and with it the renormalised state without its slot —
             * x' = ent k + B, k = 0 after a refill — plus the previous state's trip into the trail.  Depth 4 + lookup
             * (v_mad_u32_u24, v_mul_hi, v_lshl_add, v_mad_i32_i24) where it was 6; 16 instructions where there were 10. */
            uint4 nxt[8];
#pragma unroll
            for (int u = 0; u < 8; u++)
                nxt[u] = ops[u];
            /* the state between two steps, in pieces: araw + ent = the state the next symbol meets; B, k1 = its renormalised form */
            uint32_t araw = state, ent = 0, k1 = 0, B = rans_renorm(state, nxt[0].w);
            for (int k = 0; k < 64; k += 8) {
                uint4 blk[9];
#pragma unroll
                for (int u = 0; u < 8; u++)
                    blk[u] = nxt[u];
                if (k + 8 < 64) {
#pragma unroll
                    for (int u = 0; u < 8; u++)
                        nxt[u] = ops[k + 8 + u];
                }
                blk[8] = nxt[0]; /* (the chunk's last step: unused) */
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const uint4 o = blk[u];
                    uint32_t x, q, t, sd, entn, arawn, Bn, k1n;
                    asm volatile("v_mad_u32_u24 %[x], %[ent], %[k1], %[B]\n\t"
                                 "v_mul_hi_u32 %[q], %[x], %[mg]\n\t"
                                 "v_lshl_add_u32 %[x], %[x], 1, %[adr]\n\t"
                                 "v_mad_i32_i24 %[x], %[q], %[n2f], %[x]\n\t"  /* adr + 2 (x - q f) */
                                 "ds_read_u16 %[entn], %[x]\n\t"
                                 "v_add_u32 %[sd], %[araw], %[ent]\n\t"       /* the state this symbol met ... */
                                 "v_sub_u32 %[t], %[adr], %[n2f]\n\t"         /* adr + 2f */
                                 "v_mov_b32_dpp %[sd], %[tr] wave_shr:1 row_mask:0xf bank_mask:0xf\n\t" /* ... into the trail: lane 0 keeps it, lane l takes trail[l-1] */
                                 "v_cmp_ge_u32 vcc, %[x], %[t]\n\t"           /* remainder estimate >= f: the quotient is one short */
                                 "v_addc_co_u32 %[t], vcc, 0, %[q], vcc\n\t"
                                 "v_lshlrev_b32 %[t], 12, %[t]\n\t"           /* the new state without its slot */
                                 "v_lshlrev_b32 %[arawn], 12, %[q]\n\t"       /* ... and as the table will complete it (the entry carries the repair) */
                                 "v_cmp_gt_u32 vcc, %[t], %[thrn]\n\t"        /* the next symbol's refill test (entropy.c:1092) */
                                 "v_cndmask_b32_sdwa %[Bn], %[arawn], %[t], vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
                                 "v_cndmask_b32_e64 %[k1n], 1, 0, vcc\n\t"
                                 "s_waitcnt lgkmcnt(0)"
                                 : [x] "=&v"(x), [q] "=&v"(q), [t] "=&v"(t), [sd] "=&v"(sd), [entn] "=&v"(entn), [arawn] "=&v"(arawn),
                                   [Bn] "=&v"(Bn), [k1n] "=&v"(k1n)
                                 : [ent] "v"(ent), [k1] "v"(k1), [B] "v"(B), [araw] "v"(araw), [tr] "v"(trail), [mg] "v"(o.y),
                                   [adr] "v"(o.z), [n2f] "v"(o.x), [thrn] "v"(blk[u + 1].w)
                                 : "vcc");
                    trail = sd;
                    araw = arawn;
                    ent = entn;
                    B = Bn;
                    k1 = k1n;
                }
            }
            state = araw + ent;
        } else {
            for (int k = 0; k < cnt; k++) {
                const uint4 o1 = ops[k];
                HYDK_RANS_STEP(o1);
            }
        }
        __builtin_amdgcn_wave_barrier();
        /* the state seen by step j now sits in lane cnt-1-j; give it back to lane j */
        const uint32_t seen = (uint32_t)__shfl((int)trail, (cnt - 1 - lane) & 63);
        const bool refill = valid && seen > thr;
        if (DEFER) {
            /* the bit writer is another kernel (k_rans_emit, wave-parallel, as for the lane form): leave it the 16 bits
             * each symbol's refill would send and one flag per symbol, in token order */
            if (valid)
                aux[p] = (uint16_t)seen;
            const unsigned long long by_lane = __ballot(refill);
            /* lane l holds symbol (p mod 64) = cnt - 1 - l */
            const unsigned long long by_pos = __builtin_bitreverse64(by_lane) >> (64 - cnt);
            if (lane < 4 && lane * 16 < cnt)
                flags[(hi_p >> 6) * 4 + lane] = (uint16_t)(by_pos >> (16 * lane));
            refills += (uint32_t)__popcll(by_lane);
        } else {
            /* refill p is written just before residue p (entropy.c:1134-1147), i.e. prepended after it */
            const uint32_t rbits = valid ? (lo >> 16) & 0x3Fu : 0u;
            const unsigned long long residue = rec.y;
            const unsigned long long val = refill ? (residue << 16) | (seen & 0xFFFFu) : residue;
            emit(val, rbits + (refill ? 16u : 0u));
        }
    }
    if (DEFER) {
        if (lane == 0) {
            final_state_all[G] = state;
            /* [preset id][final state][per symbol: refill word, residue bits] (encoder.c:945, entropy.c:1127-1147) */
            group_bits_all[G] = (uint32_t)preset_bits + (n > 0 ? 32u : 0u) + 16u * refills + HYDK_GLOBAL(const uint32_t, jobs[slot].rbits_total)[g];
        }
        return;
    }
    /* [preset id][final state, low half first] precede everything (encoder.c:945, entropy.c:1127-1130) */
    {
        unsigned long long val = 0;
        uint32_t nb = 0;
        if (lane == 0 && n > 0) {
            val = state;
            nb = 32;
        } else if (lane == 1) {
            val = jobs[slot].preset;
            nb = (uint32_t)preset_bits;
        }
        emit(val, nb);
    }
    if (lane == 0) {
        if (cur & 31u)
            W[cur >> 5] = carry;
        group_bits_all[G] = bit_pitch_words * 32u - cur;
    }
}
#undef HYDK_RANS_STEP

/* ==========================================================================================
 * K3a, throughput form: one LANE per group.  A wavefront carries the 64 chains of one LF group, so
 * every instruction of the serial walk advances 64 groups and the entropy stage of a frame is a
 * handful of wavefronts (one per LF group) instead of hundreds: ~0.3 instructions per symbol where
 * the row forms need 4.5.  The chain only records what the bit writer needs — the 16 bits each
 * refill would send and a flag — and k_rans_emit turns records into bits, wave-parallel.
 *
 * Memory shape: every lane streams its own group.  A round is 16 symbols, aligned to the START of
 * the group's token array, so a lane reads one aligned 64-byte line of 4-byte records per round and
 * writes one 32-byte sector of refill words (its 16 u16) — whole sectors, no transposition.  The
 * round that holds the end of the stream is partial and comes first for every lane (lanes start
 * together from their own ends and finish at different times).
 * Integer sample formats only (4-byte records); float frames use k_rans_encode.
 * grid = LF groups, block = 64.
 * ======================================================================================== */

constexpr bool rec32_cluster_is_exact() {
    for (uint32_t s = 0; s < 2048u; s++)
        if (HYDK_REC32_CLUSTER(s << 4) != s / HYDK_REC32_TOKENS)
            return false;
    return true;
}
static_assert(rec32_cluster_is_exact(), "HYDK_REC32_CLUSTER: symbol / 40 by multiplication");
constexpr int kLaneTokens = (int)HYDK_REC32_TOKENS; /* integer formats: the record's symbol is cluster * 40 + token */

/* The step, and what bounds it (round 5; profiles/r05_chain_anatomy.txt).  With x the renormalised state, f the symbol's
 * frequency:  q = mulhi(x, floor(2^32 / f))  (floor(x / f) or one less),  r0 = x - q f in [0, 2f),  fix = r0 >= f,
 *   state' = ((q + fix) << 12) | slot(symbol, r0 - fix f)                                   (entropy.c:1092-1119)
 * A wavefront that has its SIMD to itself issues one instruction every ~6.7 cycles whether or not it depends on the one
 * before (scripts/ubench/valu_rate: 2.8 ns per instruction at one wave per SIMD; 9.5 cycles when dependent): the walk's
 * time is its INSTRUCTION COUNT times that, plus whatever of the slot lookup's ~64 cycles of LDS latency no instruction
 * covers.  (Measured both ways: a step cut to four dependent instructions between lookups by forming 64-bit partial
 * products ahead — 27 instructions — ran 8 % faster than round 4's 16 instructions in one dependent chain; the same
 * idea in the wave-per-group form, 9 -> 19 instructions, ran 30 % SLOWER.)  So the step below is the short one, with the
 * instructions that do not need the slot placed behind the lookup's request:
 *   - the refill test state' > (f' << 20) - 1 looks at bits 20.. of state' only: A = (q + fix) << 12 decides it, and
 *     with it x' = (A >> 16) or (A | slot) = B | (slot & mask): quotient repair, A, test, B, mask and the flag go first;
 *   - so do putting the previous state together (A | slot of the step before) and filing its low half;
 *   - a record's bits 4-14 are its operand row's byte offset (hydk_common.h): one instruction per row request.
 * A step is 17 instructions, 7 of them between the slot's arrival and the next request.  One asm statement per step:
 * the compiler pads every asm statement's edges with s_nop (it cannot see inside), three or four per step before. */
struct RansOps { /* per (cluster, token), staged in LDS: everything a step needs besides the state */
    uint32_t thr;   /* (f << 20) - 1: renormalise when state > thr (entropy.c:1092) */
    uint32_t magic; /* floor(2^32 / f) */
    uint32_t negf;  /* -f */
    uint32_t tab;   /* LDS byte address of the symbol's slot list (u16 per remainder) */
};

/* the slot table: in LDS (ds_read_u16, the step's wait is for the LDS counter) or where the table kernel left it
 * (HYDK_LANE_TAB_GLOBAL: global_load_ushort with the table's base in a scalar pair and the row's byte offset in tab; the
 * step's wait is for the vector-memory counter — loads return in order, so it also waits for whatever line of records was
 * requested before the lookup) */
#if HYDK_LANE_TAB_GLOBAL
#define HYDK_LANE_LOOKUP "global_load_ushort %[sln], %[t1], %[gb]\n\t"
#define HYDK_LANE_WAIT "s_waitcnt vmcnt(0)"
#define HYDK_LANE_GB , [gb] "s"(gtab)
#else
#define HYDK_LANE_LOOKUP "ds_read_u16 %[sln], %[t1]\n\t"
#define HYDK_LANE_WAIT "s_waitcnt lgkmcnt(0)"
#define HYDK_LANE_GB
#endif
/* request of the slot of (symbol with operands mg / nf / tab, renormalised state x) */
#define HYDK_LANE_ASM_CORE                                                                                       \
    "v_mul_hi_u32 %[q], %[x], %[mg]\n\t"                                                                         \
    "v_mad_i32_i24 %[t0], %[q], %[nf], %[x]\n\t" /* r0 = x - q f */                                              \
    "v_add_co_u32 %[t1], vcc, %[t0], %[nf]\n\t"  /* r0 - f; carry: r0 >= f, the quotient is one short */          \
    "v_min_u32 %[t0], %[t0], %[t1]\n\t"          /* x mod f */                                                    \
    "v_lshl_add_u32 %[t1], %[t0], 1, %[tab]\n\t"                                                                 \
    HYDK_LANE_LOOKUP
/* behind the request: the repaired quotient, the new state without its slot, the NEXT symbol's refill test on it */
#define HYDK_LANE_ASM_SHADOW                                                                                     \
    "v_addc_co_u32 %[q], vcc, 0, %[q], vcc\n\t"                                                                  \
    "v_lshlrev_b32 %[A], 12, %[q]\n\t"                                                                           \
    "v_cmp_gt_u32 vcc, %[A], %[thrn]\n\t"                                                                        \
    "v_cndmask_b32_sdwa %[B], %[A], %[A], vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t" \
    "v_cndmask_b32_e64 %[sm], %[k0fff], 0, vcc\n\t"                                                              \
    "v_addc_co_u32 %[fl], vcc, %[fl], %[fl], vcc\n\t"                                                            \
    HYDK_LANE_WAIT
#define HYDK_LANE_ASM_SDWA_SELECT "dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"
/* round 6's step (HYDK_LANE_STEP 2).  Between two steps of a round the walk carries A = (q + fix) << 12 (the state without
 * its slot), sl = the slot (arriving from LDS) and rf = the NEXT symbol's refill decision as a lane mask in a scalar register
 * pair (state' > thr' looks at bits 20.. only: A decides it, entropy.c:1092).  A step: st = A | sl (the state this symbol
 * meets: also what is filed), x = rf ? st >> 16 : st, then the request of its slot; behind the request the repaired
 * quotient, the new A, the next decision and its flag.  v_cndmask_b32_sdwa takes its condition from VCC only, so the pair
 * is moved there by the scalar unit (no vector issue slot); v_cmp and v_addc in their 64-bit encodings name the pair. */
#define HYDK_LANE2_TAKE                                                                                          \
    "s_mov_b64 vcc, %[rf]\n\t"                                                                                   \
    "v_or_b32 %[st], %[Ai], %[sl]\n\t"                                                                           \
    "v_cndmask_b32_sdwa %[x], %[st], %[st], vcc " HYDK_LANE_ASM_SDWA_SELECT "\n\t"
#define HYDK_LANE2_SHADOW                                                                                        \
    "v_addc_co_u32 %[q], vcc, 0, %[q], vcc\n\t"                                                                  \
    "v_lshlrev_b32 %[A], 12, %[q]\n\t"                                                                           \
    "v_cmp_gt_u32_e64 %[rfn], %[A], %[thrn]\n\t"                                                                 \
    "v_addc_co_u32_e64 %[fl], %[junk], %[fl], %[fl], %[rfn]\n\t"                                                 \
    HYDK_LANE_WAIT

/* sixteen bytes of LDS by absolute address */
__device__ __forceinline__ uint4 lds_row_at(uint32_t addr) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = *(__attribute__((address_space(3))) const u32x4 *)(uintptr_t)addr;
    return uint4{v.x, v.y, v.z, v.w};
}
/* LDS of a lane-form chain workgroup: operand rows + slot tables (or the LF code builder's scratch, whichever is larger) */
constexpr int lanes_lds_bytes(int nc) {
    const int ops = nc * kLaneTokens * (int)sizeof(uint4), tab = HYDK_LANE_TAB_GLOBAL ? 0 : 2 * nc * HYDK_ANS_SLOTS;
    const int need = ops + tab > (int)sizeof(LfHuffScratch) ? ops + tab : (int)sizeof(LfHuffScratch);
    return nc == 9 && need < HYDK_CHAIN_LDS_MIN ? HYDK_CHAIN_LDS_MIN : need;
}
template <int NC> /* NC: clusters per preset of the frame's clustering scheme (9 / 3 / 2 / 1): the tables' size in LDS */
__global__ __launch_bounds__(64) void k_rans_lanes(const HydkLfJob *__restrict__ jobs, const uint32_t *sym_count_all,
                                                   const HydkTables *tabs, uint16_t *aux_all, uint16_t *flags_all,
                                                   uint32_t aux_pitch /* symbols per group in aux / flags */,
                                                   uint32_t *final_state_all, uint32_t *group_bits_all,
                                                   int preset_bits, const uint32_t *status, int num_slots,
                                                   const uint32_t *lf_hist, HydkLfStream *lf_streams, void *lf_work) {
    /* The workgroup HOLDS its LDS for the whole walk, and in the pipelined loop that (times the walk's duration, which
     * more than doubles beside other frames' transform workgroups) is what the chains cost the kernels around them
     * (profiles/r04_pipeline_bounds.txt, r05_pipeline_bounds.txt).  The tables are sized by the frame's clustering scheme
     * (a 16384^2 frame has 3 clusters per preset: 26 KB, not 80). */
    constexpr int kOpsBytes = NC * kLaneTokens * (int)sizeof(uint4);
    constexpr int kTabBytes = HYDK_LANE_TAB_GLOBAL ? 0 : 2 * NC * HYDK_ANS_SLOTS;
    constexpr int kNeedBytes = kOpsBytes + kTabBytes > (int)sizeof(LfHuffScratch) ? kOpsBytes + kTabBytes : (int)sizeof(LfHuffScratch);
    /* HYDK_CHAIN_LDS_MIN (round 6): a nine-cluster chain workgroup asks for at least this many bytes.  79.5 KB is just UNDER half
     * of a compute unit's 160 KB: two chain workgroups fit one compute unit and then no transform workgroup does (2 x 63 of 128
     * granules) — that unit's vector pipelines idle for a chain's lifetime.  Above half (65 granules = 83 200 bytes) a compute unit
     * holds one chain at most and always two transform workgroups beside it (65 + 2 x 25 = 115). */
    constexpr int kLdsBytes = NC == 9 && kNeedBytes < HYDK_CHAIN_LDS_MIN ? HYDK_CHAIN_LDS_MIN : kNeedBytes;
    static_assert(kLdsBytes == lanes_lds_bytes(NC), "the launch asks for what the kernel lays out");
#if HYDK_CHAIN_DYN_LDS
    /* the SAME bytes, asked for at launch: a kernel whose STATIC LDS lets only two of its workgroups onto a compute unit has its
     * register allocation padded by the compiler to the smallest figure that rules out a second wavefront per SIMD
     * (.amdhsa_next_free_vgpr 257 for a kernel that uses 164) — registers no wavefront of ANOTHER kernel can then have */
    extern __shared__ __attribute__((aligned(16))) unsigned char s_mem[];
#else
    __shared__ __attribute__((aligned(16))) unsigned char s_mem[kLdsBytes];
#endif
    uint4 *const s_ops = (uint4 *)s_mem;
    unsigned char *const s_tab = s_mem + kOpsBytes; /* uint16_t[NC * 4096] */
    __builtin_amdgcn_s_setprio(HYDK_CHAIN_PRIO);
#if HYDK_CHAIN_DYN_LDS
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)s_mem != 0u)
        __builtin_trap(); /* (HYDK_LANE_ROW addresses the rows absolutely) */
#endif
#if HYDK_CHAIN_PROBE & 32
    const uint32_t probe_t0 = (uint32_t)__builtin_amdgcn_s_memrealtime();
#endif
#if HYDK_CHAIN_HOG
    asm volatile("v_accvgpr_write_b32 a255, 0" ::: "a255");
#endif
    const int lane = threadIdx.x;
    if ((int)blockIdx.x >= num_slots) {
        /* passengers: workgroup num_slots + s builds the prefix code of LF group s's coefficient stream
         * (lf_coder.hip; 200 us of serial work on one wavefront, like the chains and independent of them) */
        const int s = (int)blockIdx.x - num_slots;
        lf_huffman_wave(lf_hist + (size_t)s * HYDK_LF_CODES, ((LfWork *)lf_work)[s].codes, lf_streams + s, *(LfHuffScratch *)s_mem, lane);
        return;
    }
    const int slot = blockIdx.x;
    const int ngroups = jobs[slot].gcols * jobs[slot].grows;
    const HydkTables *tab = tabs + slot;
    if (*status & HYDK_STATUS_OVERFLOW)
        return; /* the transform stage ran out of token space: the host reruns the frame */
    typedef __attribute__((address_space(3))) const uint16_t LdsU16;
    {
#if HYDK_LANE_TAB_GLOBAL
        const uint32_t tab_lds = 0; /* a row's byte offset from the table's base, which the lookups take from a scalar pair */
#else
        const uint4 *src = (const uint4 *)&tab->inv1[0][0];
        for (int i = lane; i < NC * HYDK_ANS_SLOTS / 8; i += 64)
            ((uint4 *)s_tab)[i] = src[i];
        const uint32_t tab_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)s_tab;
#endif
        for (int i = lane; i < NC * kLaneTokens; i += 64) {
            const int c = i / kLaneTokens, at = c * HYDK_ALPHABET + i % kLaneTokens;
            const uint32_t fbv = (&tab->fb[0][0])[at], f = fbv & 0xFFFFu;
            uint4 o;
            o.x = f ? (f << 20) - 1u : 0xFFFFFFFFu;
            o.y = (&tab->magic[0][0])[at];
            o.z = 0u - f;
            o.w = tab_lds + 2u * ((uint32_t)c * HYDK_ANS_SLOTS + (fbv >> 16));
            s_ops[i] = o;
        }
    }
    __syncthreads();
#if HYDK_LANE_TAB_GLOBAL
    /* wave-uniform: lives in a scalar register pair */
    const unsigned long long gtab = (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)&tab->inv1[0][0]) |
                                    (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uintptr_t)&tab->inv1[0][0] >> 32)) << 32;
#endif
    const size_t G = (size_t)slot * HYDK_GROUPS_PER_LFG + lane;
    const int n = lane < ngroups ? (int)sym_count_all[G] : 0;
    /* 4-byte records, 4 per uint4 (tok_cap is a multiple of 16: rounds never straddle a group's array) */
    const void *tok = (const char *)jobs[slot].tokens + (size_t)lane * jobs[slot].tok_cap * jobs[slot].rec_bytes;
    uint4 *aux = (uint4 *)(aux_all + G * aux_pitch); /* u16 per symbol, 8 per uint4 */
    uint16_t *flags = flags_all + G * (aux_pitch / 16);
    uint32_t refills = 0;
    int rj = ((n + 15) >> 4) - 1; /* this lane's current round = 16-symbol chunk index; -1: nothing (left) to do */
    int rounds = rj + 1;
#pragma unroll
    for (int d = 32; d; d >>= 1)
        rounds = max(rounds, __shfl_xor(rounds, d));
    rounds = __builtin_amdgcn_readfirstlane(rounds);
    const int first_count = n - 16 * rj; /* symbols in the lane's first (partial) round: 1..16 */

    uint32_t state, k0fff, ksel;
    asm volatile("v_mov_b32 %0, 0x130000" : "=v"(state));
    asm volatile("v_mov_b32 %0, 0xfff" : "=v"(k0fff));       /* constants the walk wants in vector registers */
    asm volatile("v_mov_b32 %0, 0x05040100" : "=v"(ksel));   /* v_perm_b32: {low half of source 0, low half of source 1} */
    uint4 nx[4];
#pragma unroll
    for (int q = 0; q < 4; q++)
        nx[q] = rj >= 0 ? load_records4(tok, rj * 4 + q) : uint4{0, 0, 0, 0};

/* a position beyond the stream's end walks with row 0's operands, whose symbol may never have occurred (f = 0: the "remainder"
 * is the state itself): an LDS read past the end returns 0, a global one must not be made */
/* an operand row by its byte offset (bits 4-14 of a record).  With the LDS asked for at launch the compiler adds the array's
 * (link-time) base to every request — one more vector instruction per symbol; the kernel has no other LDS, the array sits at
 * address 0 (checked at the kernel's start), and the offset IS the address */
#if HYDK_CHAIN_DYN_LDS
#define HYDK_LANE_ROW(row) lds_row_at((uint32_t)(row))
#else
#define HYDK_LANE_ROW(row) (*(const uint4 *)(s_mem + (row)))
#endif
#if HYDK_LANE_TAB_GLOBAL
#define HYDK_LANE_COLD_SLOT(off, VALID) ((uint32_t) * HYDK_GLOBAL(const uint16_t, (const char *)(uintptr_t)gtab + ((VALID) ? (off) : 0u)))
#else
#define HYDK_LANE_COLD_SLOT(off, VALID) ((uint32_t) * (LdsU16 *)(uintptr_t)(off))
#endif
/* one symbol the plain way (the first, partial round of a lane): the step only counts if VALID */
#define HYDK_LANE_STEP_COLD(o, pos, VALID)                                                                       \
    do {                                                                                                         \
        uint32_t x;                                                                                              \
        const uint32_t thr = (VALID) ? o.x : 0xFFFFFFFFu;                                                        \
        asm("v_cmp_gt_u32 vcc, %2, %3\n\t"                                                                       \
            "v_cndmask_b32_sdwa %0, %2, %2, vcc " HYDK_LANE_ASM_SDWA_SELECT "\n\t"                               \
            "v_addc_co_u32 %1, vcc, %1, %1, vcc"                                                                 \
            : "=&v"(x), "+v"(fl)                                                                                 \
            : "v"(state), "v"(thr)                                                                               \
            : "vcc");                                                                                            \
        if ((pos) & 1) /* the walk goes down: the odd position of a pair comes first */                          \
            w16[(pos) >> 1] = state << 16;                                                                       \
        else                                                                                                     \
            w16[(pos) >> 1] |= state & 0xFFFFu;                                                                  \
        uint32_t q = __umulhi(x, o.y);                                                                           \
        const uint32_t r0 = mad24(q, o.z, x);                                                                    \
        uint32_t r1, q1;                                                                                         \
        asm("v_add_co_u32 %0, vcc, %2, %3\n\t"                                                                   \
            "v_addc_co_u32 %1, vcc, 0, %4, vcc"                                                                  \
            : "=&v"(r1), "=v"(q1)                                                                                \
            : "v"(r0), "v"(o.z), "v"(q)                                                                          \
            : "vcc");                                                                                            \
        const uint32_t nstate = (q1 << 12) | HYDK_LANE_COLD_SLOT(o.w + 2u * min(r0, r1), VALID);                \
        state = (VALID) ? nstate : state;                                                                        \
    } while (0)

#if HYDK_LANE_STEP == 1
/* a round's first symbol (position 15): the state arrives in one piece */
#define HYDK_LANE_STEP_HEAD(o, on)                                                                               \
    do {                                                                                                         \
        uint32_t x, q, t0, t1, sln;                                                                              \
        asm volatile("v_cmp_gt_u32 vcc, %[st], %[thr]\n\t"                                                       \
                     "v_cndmask_b32_sdwa %[x], %[st], %[st], vcc " HYDK_LANE_ASM_SDWA_SELECT "\n\t"              \
                     "v_addc_co_u32 %[fl], vcc, %[fl], %[fl], vcc\n\t" HYDK_LANE_ASM_CORE HYDK_LANE_ASM_SHADOW   \
                     : [x] "=&v"(x), [q] "=&v"(q), [t0] "=&v"(t0), [t1] "=&v"(t1), [sln] "=&v"(sln), [A] "=&v"(A), \
                       [B] "=&v"(B), [sm] "=&v"(sm), [fl] "+v"(fl)                                               \
                     : [st] "v"(state), [thr] "v"(o.x), [mg] "v"(o.y), [nf] "v"(o.z), [tab] "v"(o.w) HYDK_LANE_GB, [thrn] "v"(on.x), \
                       [k0fff] "v"(k0fff)                                                                        \
                     : "vcc");                                                                                   \
        so = state; /* position 15 is odd: filed together with position 14's */                                  \
        sl = sln;                                                                                                \
    } while (0)
/* a symbol in the middle of a round: (A, sl) = the state it meets, in two pieces; B, sm = the renormalised state's */
#define HYDK_LANE_STEP_BODY(o, on, pos)                                                                          \
    do {                                                                                                         \
        uint32_t x, q, t0, t1, sln, st, An, Bn, smn;                                                             \
        if ((pos) & 1) {                                                                                         \
            asm volatile("v_and_or_b32 %[x], %[sl], %[smi], %[Bi]\n\t" HYDK_LANE_ASM_CORE                        \
                         "v_or_b32 %[st], %[Ai], %[sl]\n\t" HYDK_LANE_ASM_SHADOW                                 \
                         : [x] "=&v"(x), [q] "=&v"(q), [t0] "=&v"(t0), [t1] "=&v"(t1), [sln] "=&v"(sln), [st] "=&v"(st), \
                           [A] "=&v"(An), [B] "=&v"(Bn), [sm] "=&v"(smn), [fl] "+v"(fl)                          \
                         : [sl] "v"(sl), [smi] "v"(sm), [Bi] "v"(B), [Ai] "v"(A), [mg] "v"(o.y), [nf] "v"(o.z),   \
                           [tab] "v"(o.w) HYDK_LANE_GB, [thrn] "v"(on.x), [k0fff] "v"(k0fff)                                  \
                         : "vcc");                                                                               \
            so = st;                                                                                             \
        } else {                                                                                                 \
            asm volatile("v_and_or_b32 %[x], %[sl], %[smi], %[Bi]\n\t" HYDK_LANE_ASM_CORE                        \
                         "v_or_b32 %[st], %[Ai], %[sl]\n\t"                                                      \
                         "v_perm_b32 %[w], %[so], %[st], %[ksel]\n\t" HYDK_LANE_ASM_SHADOW                       \
                         : [x] "=&v"(x), [q] "=&v"(q), [t0] "=&v"(t0), [t1] "=&v"(t1), [sln] "=&v"(sln), [st] "=&v"(st), \
                           [w] "=&v"(w16[(pos) >> 1]), [A] "=&v"(An), [B] "=&v"(Bn), [sm] "=&v"(smn), [fl] "+v"(fl) \
                         : [sl] "v"(sl), [smi] "v"(sm), [Bi] "v"(B), [Ai] "v"(A), [so] "v"(so), [ksel] "v"(ksel), \
                           [mg] "v"(o.y), [nf] "v"(o.z), [tab] "v"(o.w) HYDK_LANE_GB, [thrn] "v"(on.x), [k0fff] "v"(k0fff)    \
                         : "vcc");                                                                               \
        }                                                                                                        \
        A = An;                                                                                                  \
        B = Bn;                                                                                                  \
        sm = smn;                                                                                                \
        sl = sln;                                                                                                \
    } while (0)
/* a round's last symbol (position 0): the round ends with the state in one piece */
#define HYDK_LANE_STEP_TAIL(o)                                                                                   \
    do {                                                                                                         \
        uint32_t x, q, t0, t1, sln, st, An;                                                                      \
        asm volatile("v_and_or_b32 %[x], %[sl], %[smi], %[Bi]\n\t" HYDK_LANE_ASM_CORE                            \
                     "v_or_b32 %[st], %[Ai], %[sl]\n\t"                                                          \
                     "v_perm_b32 %[w], %[so], %[st], %[ksel]\n\t"                                                \
                     "v_addc_co_u32 %[q], vcc, 0, %[q], vcc\n\t"                                                 \
                     "v_lshlrev_b32 %[A], 12, %[q]\n\t"                                                          \
                     HYDK_LANE_WAIT "\n\t"                                                                       \
                     "v_or_b32 %[state], %[A], %[sln]"                                                           \
                     : [x] "=&v"(x), [q] "=&v"(q), [t0] "=&v"(t0), [t1] "=&v"(t1), [sln] "=&v"(sln), [st] "=&v"(st), \
                       [w] "=&v"(w16[0]), [A] "=&v"(An), [state] "=&v"(state)                                    \
                     : [sl] "v"(sl), [smi] "v"(sm), [Bi] "v"(B), [Ai] "v"(A), [so] "v"(so), [ksel] "v"(ksel),     \
                       [mg] "v"(o.y), [nf] "v"(o.z), [tab] "v"(o.w) HYDK_LANE_GB                                              \
                     : "vcc");                                                                                   \
    } while (0)

#else
/* a round's first symbol (position 15): the state arrives in one piece */
#define HYDK_LANE_STEP_HEAD(o, on)                                                                               \
    do {                                                                                                         \
        uint32_t x, q, t0, t1, sln;                                                                              \
        unsigned long long rfn, junk;                                                                            \
        asm volatile("v_cmp_gt_u32 vcc, %[st], %[thr]\n\t"                                                       \
                     "v_cndmask_b32_sdwa %[x], %[st], %[st], vcc " HYDK_LANE_ASM_SDWA_SELECT "\n\t"              \
                     "v_addc_co_u32 %[fl], vcc, %[fl], %[fl], vcc\n\t" HYDK_LANE_ASM_CORE HYDK_LANE2_SHADOW      \
                     : [x] "=&v"(x), [q] "=&v"(q), [t0] "=&v"(t0), [t1] "=&v"(t1), [sln] "=&v"(sln), [A] "=&v"(A), \
                       [rfn] "=&s"(rfn), [junk] "=&s"(junk), [fl] "+v"(fl)                                       \
                     : [st] "v"(state), [thr] "v"(o.x), [mg] "v"(o.y), [nf] "v"(o.z), [tab] "v"(o.w) HYDK_LANE_GB, [thrn] "v"(on.x) \
                     : "vcc");                                                                                   \
        so = state; /* position 15 is odd: filed together with position 14's */                                  \
        sl = sln;                                                                                                \
        rf = rfn;                                                                                                \
    } while (0)
/* a symbol in the middle of a round */
#define HYDK_LANE_STEP_BODY(o, on, pos)                                                                          \
    do {                                                                                                         \
        uint32_t x, q, t0, t1, sln, st, An;                                                                      \
        unsigned long long rfn, junk;                                                                            \
        if ((pos) & 1) {                                                                                         \
            asm volatile(HYDK_LANE2_TAKE HYDK_LANE_ASM_CORE HYDK_LANE2_SHADOW                                    \
                         : [x] "=&v"(x), [q] "=&v"(q), [t0] "=&v"(t0), [t1] "=&v"(t1), [sln] "=&v"(sln), [st] "=&v"(st), \
                           [A] "=&v"(An), [rfn] "=&s"(rfn), [junk] "=&s"(junk), [fl] "+v"(fl)                    \
                         : [sl] "v"(sl), [Ai] "v"(A), [rf] "s"(rf), [mg] "v"(o.y), [nf] "v"(o.z), [tab] "v"(o.w) HYDK_LANE_GB, \
                           [thrn] "v"(on.x)                                                                      \
                         : "vcc");                                                                               \
            so = st;                                                                                             \
        } else {                                                                                                 \
            asm volatile(HYDK_LANE2_TAKE HYDK_LANE_ASM_CORE                                                      \
                         "v_perm_b32 %[w], %[so], %[st], %[ksel]\n\t" HYDK_LANE2_SHADOW                          \
                         : [x] "=&v"(x), [q] "=&v"(q), [t0] "=&v"(t0), [t1] "=&v"(t1), [sln] "=&v"(sln), [st] "=&v"(st), \
                           [w] "=&v"(w16[(pos) >> 1]), [A] "=&v"(An), [rfn] "=&s"(rfn), [junk] "=&s"(junk), [fl] "+v"(fl) \
                         : [sl] "v"(sl), [Ai] "v"(A), [rf] "s"(rf), [so] "v"(so), [ksel] "v"(ksel), [mg] "v"(o.y), \
                           [nf] "v"(o.z), [tab] "v"(o.w) HYDK_LANE_GB, [thrn] "v"(on.x)                                       \
                         : "vcc");                                                                               \
        }                                                                                                        \
        A = An;                                                                                                  \
        rf = rfn;                                                                                                \
        sl = sln;                                                                                                \
    } while (0)
/* a round's last symbol (position 0): the round ends with the state in one piece */
#define HYDK_LANE_STEP_TAIL(o)                                                                                   \
    do {                                                                                                         \
        uint32_t x, q, t0, t1, sln, st, An;                                                                      \
        asm volatile(HYDK_LANE2_TAKE HYDK_LANE_ASM_CORE                                                          \
                     "v_perm_b32 %[w], %[so], %[st], %[ksel]\n\t"                                                \
                     "v_addc_co_u32 %[q], vcc, 0, %[q], vcc\n\t"                                                 \
                     "v_lshlrev_b32 %[A], 12, %[q]\n\t"                                                          \
                     HYDK_LANE_WAIT "\n\t"                                                                       \
                     "v_or_b32 %[state], %[A], %[sln]"                                                           \
                     : [x] "=&v"(x), [q] "=&v"(q), [t0] "=&v"(t0), [t1] "=&v"(t1), [sln] "=&v"(sln), [st] "=&v"(st), \
                       [w] "=&v"(w16[0]), [A] "=&v"(An), [state] "=&v"(state)                                    \
                     : [sl] "v"(sl), [Ai] "v"(A), [rf] "s"(rf), [so] "v"(so), [ksel] "v"(ksel), [mg] "v"(o.y),    \
                       [nf] "v"(o.z), [tab] "v"(o.w) HYDK_LANE_GB                                                             \
                     : "vcc");                                                                                   \
    } while (0)
#endif /* HYDK_LANE_STEP */

#if HYDK_LANE_PIPE == 0
/* one round = the 16 symbols of one 64-byte line of records.  The first round of a lane is the partial one (FIRST: only
 * positions below first_count count); it is peeled out of the loop and walks the plain way */
#define HYDK_LANE_ROUND(FIRST)                                                                                   \
    do {                                                                                                         \
        uint4 cur[4];                                                                                            \
        _Pragma("unroll") for (int q = 0; q < 4; q++) cur[q] = nx[q];                                            \
        const int rjn = rj - 1;                                                                                  \
        if (rjn >= 0 && !(HYDK_CHAIN_PROBE & 4)) { /* the next round's line travels during this round's walk */  \
            _Pragma("unroll") for (int q = 0; q < 4; q++) nx[q] = load_records4(tok, rjn * 4 + q);               \
        }                                                                                                        \
        if (rj >= 0) {                                                                                           \
            uint32_t fl = 0;                                                                                     \
            uint32_t w16[8];                                                                                     \
            const uint32_t recs[16] = {cur[0].x, cur[0].y, cur[0].z, cur[0].w, cur[1].x, cur[1].y, cur[1].z, cur[1].w, \
                                       cur[2].x, cur[2].y, cur[2].z, cur[2].w, cur[3].x, cur[3].y, cur[3].z, cur[3].w}; \
            /* bits 4-14 of a record are its operand row's byte offset (beyond the stream's end: stale bytes; an LDS \
             * read past the table returns 0).  All sixteen rows are requested before the walk: a row requested  \
             * inside its step returns behind the step's slot lookup and lengthens every wait */                 \
            uint4 ov[16];                                                                                        \
            _Pragma("unroll") for (int pos = 15; pos >= 0; pos--) {                                              \
                uint32_t row = (FIRST) && !(pos < first_count) ? 0u : recs[pos] & 0x7FF0u;                       \
                if (HYDK_CHAIN_PROBE & 1) { /* one address for all lanes, sixteen requests all the same */       \
                    row = 1920u;                                                                                 \
                    asm volatile("" : "+v"(row));                                                                \
                }                                                                                                \
                ov[pos] = HYDK_LANE_ROW(row);                                                         \
            }                                                                                                    \
            __builtin_amdgcn_sched_barrier(0); /* (the scheduler would sink the requests back into the steps) */ \
            uint32_t A = 0, B = 0, sm = 0, sl = 0, so = 0; /* the walk's state between two steps of a round */   \
            unsigned long long rf = 0; /* (HYDK_LANE_STEP 2) the next symbol's refill decision, a lane mask */   \
            (void)B; (void)sm; (void)rf;                                                                         \
            if (FIRST) {                                                                                         \
                _Pragma("unroll") for (int pos = 15; pos >= 8; pos--)                                            \
                    HYDK_LANE_STEP_COLD(ov[pos], pos, pos < first_count);                                        \
            } else {                                                                                             \
                HYDK_LANE_STEP_HEAD(ov[15], ov[14]);                                                             \
                _Pragma("unroll") for (int pos = 14; pos >= 8; pos--)                                            \
                    HYDK_LANE_STEP_BODY(ov[pos], ov[pos - 1], pos);                                              \
            }                                                                                                    \
            /* the PREVIOUS round's refill words and flags are stored here, in the middle of the walk: vmcnt counts \
             * stores too, and stored at a round's end they were the youngest memory operations when the next round \
             * took its records — that wait was a wait for the stores' round trip */                             \
            __builtin_amdgcn_sched_barrier(0);                                                                   \
            if (prj >= 0 && !(HYDK_CHAIN_PROBE & 4)) {                                                           \
                aux[prj * 2] = uint4{pw[0], pw[1], pw[2], pw[3]};                                                \
                aux[prj * 2 + 1] = uint4{pw[4], pw[5], pw[6], pw[7]};                                            \
                flags[prj] = (uint16_t)pfl; /* bit (p mod 16): symbol p refills */                               \
            }                                                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                                   \
            if (FIRST) {                                                                                         \
                _Pragma("unroll") for (int pos = 7; pos >= 0; pos--)                                             \
                    HYDK_LANE_STEP_COLD(ov[pos], pos, pos < first_count);                                        \
            } else {                                                                                             \
                _Pragma("unroll") for (int pos = 7; pos >= 1; pos--)                                             \
                    HYDK_LANE_STEP_BODY(ov[pos], ov[pos - 1], pos);                                              \
                HYDK_LANE_STEP_TAIL(ov[0]);                                                                      \
            }                                                                                                    \
            _Pragma("unroll") for (int q = 0; q < 8; q++) pw[q] = w16[q];                                        \
            pfl = fl;                                                                                            \
            refills += (uint32_t)__popc(fl);                                                                     \
        } else if (prj >= 0) { /* this lane's last round */                                                      \
            aux[prj * 2] = uint4{pw[0], pw[1], pw[2], pw[3]};                                                    \
            aux[prj * 2 + 1] = uint4{pw[4], pw[5], pw[6], pw[7]};                                                \
            flags[prj] = (uint16_t)pfl;                                                                          \
        }                                                                                                        \
        prj = rj; /* (negative: nothing to store) */                                                             \
        rj = rjn;                                                                                                \
    } while (0)

    uint32_t pw[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pfl = 0;
    int prj = -1; /* the round whose results are still in registers */
    if (rounds > 0)
        HYDK_LANE_ROUND(true);
    for (int it = 1; it < rounds; it++)
        HYDK_LANE_ROUND(false);
#undef HYDK_LANE_ROUND
    if (prj >= 0) {
        aux[prj * 2] = uint4{pw[0], pw[1], pw[2], pw[3]};
        aux[prj * 2 + 1] = uint4{pw[4], pw[5], pw[6], pw[7]};
        flags[prj] = (uint16_t)pfl;
    }
#else
/* one round = the 16 symbols of one 64-byte line of records.  The first round of a lane is the partial one (FIRST: only
 * positions below first_count count); it is peeled out of the loop and walks the plain way.
 * Round 6 (HYDK_LANE_PIPE 1 / 2): the lines travel in two / three register buffers that take turns (the loop is unrolled
 * by that many rounds) instead of one buffer copied into another at every round's start — sixteen v_mov a round, one
 * vector issue slot per symbol, gone; the line requested in a round is the one DIST rounds ahead (inside the pipelined loop
 * a request that left one round = 2 us ago has often not returned: the transform kernels of fifteen other frames keep the
 * memory system at 2-3 TB/s; profiles/r06_loop_stage_times.txt: without global traffic a chain lasts 2.6 ms in the loop
 * instead of 4.4); and the PREVIOUS round's refill words and flags are stored at the round's start, right behind that
 * request, so that they have the whole walk to complete before the next wait for records (vmcnt counts stores too: round 5
 * stored them in the middle of the walk, eight steps before that wait). */
#define HYDK_LANE_STORE_PREV()                                                                                   \
    do {                                                                                                         \
        if (prj >= 0 && !(HYDK_CHAIN_PROBE & (4 | 8))) {                                                         \
            aux[prj * 2] = uint4{pw[0], pw[1], pw[2], pw[3]};                                                    \
            aux[prj * 2 + 1] = uint4{pw[4], pw[5], pw[6], pw[7]};                                                \
            flags[prj] = (uint16_t)pfl; /* bit (p mod 16): symbol p refills */                                   \
        }                                                                                                        \
    } while (0)
#define HYDK_LANE_ROUND(FIRST, CUR, AFTER_REQUESTS)                                                              \
    do {                                                                                                         \
        if (rj >= 0) {                                                                                           \
            uint32_t fl = 0;                                                                                     \
            uint32_t w16[8];                                                                                     \
            const uint32_t recs[16] = {CUR[0].x, CUR[0].y, CUR[0].z, CUR[0].w, CUR[1].x, CUR[1].y, CUR[1].z, CUR[1].w, \
                                       CUR[2].x, CUR[2].y, CUR[2].z, CUR[2].w, CUR[3].x, CUR[3].y, CUR[3].z, CUR[3].w}; \
            /* bits 4-14 of a record are its operand row's byte offset (beyond the stream's end: stale bytes; an LDS \
             * read past the table returns 0).  All sixteen rows are requested before the walk: a row requested  \
             * inside its step returns behind the step's slot lookup and lengthens every wait */                 \
            uint4 ov[16];                                                                                        \
            _Pragma("unroll") for (int pos = 15; pos >= 0; pos--) {                                              \
                uint32_t row = (FIRST) && !(pos < first_count) ? 0u : recs[pos] & 0x7FF0u;                       \
                if (HYDK_CHAIN_PROBE & 1) { /* one address for all lanes, sixteen requests all the same */       \
                    row = 1920u;                                                                                 \
                    asm volatile("" : "+v"(row));                                                                \
                }                                                                                                \
                ov[pos] = HYDK_LANE_ROW(row);                                                         \
            }                                                                                                    \
            __builtin_amdgcn_sched_barrier(0); /* (the scheduler would sink the requests back into the steps) */ \
            /* memory operations go out HERE, behind the wait the row requests needed for this round's records (the \
             * compiler drains the whole counter there): whatever is requested now has until the next such wait */ \
            AFTER_REQUESTS;                                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                                   \
            uint32_t A = 0, B = 0, sm = 0, sl = 0, so = 0; /* the walk's state between two steps of a round */   \
            unsigned long long rf = 0; /* (HYDK_LANE_STEP 2) the next symbol's refill decision, a lane mask */   \
            (void)B; (void)sm; (void)rf;                                                                         \
            if (FIRST) {                                                                                         \
                _Pragma("unroll") for (int pos = 15; pos >= 0; pos--)                                            \
                    HYDK_LANE_STEP_COLD(ov[pos], pos, pos < first_count);                                        \
            } else {                                                                                             \
                HYDK_LANE_STEP_HEAD(ov[15], ov[14]);                                                             \
                _Pragma("unroll") for (int pos = 14; pos >= 1; pos--)                                            \
                    HYDK_LANE_STEP_BODY(ov[pos], ov[pos - 1], pos);                                              \
                HYDK_LANE_STEP_TAIL(ov[0]);                                                                      \
            }                                                                                                    \
            _Pragma("unroll") for (int q = 0; q < 8; q++) pw[q] = w16[q];                                        \
            pfl = fl;                                                                                            \
            refills += (uint32_t)__popc(fl);                                                                     \
        } else {                                                                                                 \
            HYDK_LANE_STORE_PREV(); /* this lane's last round */                                                 \
        }                                                                                                        \
        prj = rj; /* (negative: nothing to store) */                                                             \
        rj = rj - 1;                                                                                             \
    } while (0)
/* the lines AHEAD .. AHEAD + COUNT - 1 rounds ahead of the current one, into F0 (and F1) */
#define HYDK_LANE_FETCH(F, AHEAD)                                                                                \
    do {                                                                                                         \
        if (rj - (AHEAD) >= 0 && !(HYDK_CHAIN_PROBE & (4 | 16))) {                                               \
            _Pragma("unroll") for (int q = 0; q < 4; q++) F[q] = load_records4(tok, (rj - (AHEAD)) * 4 + q);     \
        }                                                                                                        \
    } while (0)

    uint32_t pw[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pfl = 0;
    int prj = -1; /* the round whose results are still in registers */
    int it = 0;   /* rounds walked (wave-uniform, like `rounds`) */
#if HYDK_LANE_PIPE == 1
    /* two buffers: the next round's line is requested behind this round's wait (one round = sixteen steps of slack) */
    uint4 nb[4] = {};
#define HYDK_LANE_TURN(FIRST, C, F)                                                                              \
    if (it < rounds) {                                                                                           \
        __builtin_amdgcn_s_waitcnt(0x0F70); /* vmcnt(0), for every lane */                                       \
        HYDK_LANE_ROUND(FIRST, C, HYDK_LANE_FETCH(F, 1); HYDK_LANE_STORE_PREV());                                \
        it++;                                                                                                    \
    }
    HYDK_LANE_TURN(true, nx, nb)
    while (it < rounds) {
        HYDK_LANE_TURN(false, nb, nx)
        HYDK_LANE_TURN(false, nx, nb)
    }
#else
    /* four buffers in two pairs: a pair of rounds walks one pair while the other pair's two lines travel (two rounds =
     * thirty-two steps of slack between a request and the wait that needs it) */
    uint4 ny[4], nb[4] = {}, nc[4] = {};
#pragma unroll
    for (int q = 0; q < 4; q++) /* the second line travels with the first */
        ny[q] = rj >= 1 && !(HYDK_CHAIN_PROBE & 4) ? load_records4(tok, (rj - 1) * 4 + q) : uint4{0, 0, 0, 0};
#define HYDK_LANE_TURN(FIRST, C0, C1, F0, F1)                                                                    \
    if (it < rounds) {                                                                                           \
        /* the ONE wait of the pair, for every lane (a lane that has finished takes the other branch of the round: \
         * left to the compiler, that path would leave the lines "pending" and the pair's second round would drain \
         * the counter again, the lines just requested included) */                                              \
        __builtin_amdgcn_s_waitcnt(0x0F70); /* vmcnt(0) */                                                       \
        HYDK_LANE_ROUND(FIRST, C0, HYDK_LANE_FETCH(F0, 2); HYDK_LANE_FETCH(F1, 3); HYDK_LANE_STORE_PREV());      \
        it++;                                                                                                    \
        if (it < rounds) {                                                                                       \
            HYDK_LANE_ROUND(false, C1, HYDK_LANE_STORE_PREV());                                                  \
            it++;                                                                                                \
        }                                                                                                        \
    }
    HYDK_LANE_TURN(true, nx, ny, nb, nc)
    while (it < rounds) {
        HYDK_LANE_TURN(false, nb, nc, nx, ny)
        HYDK_LANE_TURN(false, nx, ny, nb, nc)
    }
#endif
#undef HYDK_LANE_TURN
#undef HYDK_LANE_FETCH
#undef HYDK_LANE_ROUND
    HYDK_LANE_STORE_PREV();
#undef HYDK_LANE_STORE_PREV
#endif /* HYDK_LANE_PIPE */
#undef HYDK_LANE_STEP_TAIL
#undef HYDK_LANE_STEP_BODY
#undef HYDK_LANE_STEP_HEAD
#undef HYDK_LANE_STEP_COLD
#undef HYDK_LANE_COLD_SLOT
#undef HYDK_LANE_ROW
    if (lane < ngroups) {
        final_state_all[G] = state;
        /* [preset id][final state][per symbol: refill word, residue bits] (encoder.c:945, entropy.c:1127-1147) */
        group_bits_all[G] = (uint32_t)preset_bits + (n > 0 ? 32u : 0u) + 16u * refills + HYDK_GLOBAL(const uint32_t, jobs[slot].rbits_total)[lane];
    } else {
        group_bits_all[G] = 0;
    }
#if HYDK_CHAIN_PROBE & 32
    /* timing only: when this wavefront started and ended (100 MHz ticks), where hydamd_read_sections finds them (lanes 0 and
     * 1 of the slot): were the chains of a launch RUNNING for the stage's duration in the loop, or WAITING for a compute unit
     * with 80 KB of LDS free (scripts/pipe_probe.py --chain-clock) */
    if (lane == 0)
        group_bits_all[G] = probe_t0;
    if (lane == 1)
        group_bits_all[G] = (uint32_t)__builtin_amdgcn_s_memrealtime();
#endif
}

/* one wave per group: records + the chain's refill words -> bits, written straight to the section's
 * place in the frame's payload (k_scan_sections has turned the chain's bit counts into byte offsets
 * and cleared the two words a section may share with its neighbours).  The section is filled from
 * its END in batches of 512 symbols aligned to the start of the token array (eight consecutive symbols
 * per lane: two aligned 16-byte record loads, one 16-byte load of refill words); inside a batch the
 * stream order [refill_p][residue_p][refill_p+1].. is plain ascending (lane, symbol) order.
 * grid = 16 x LF groups, block = 256. */
#ifndef HYDK_EMIT_SHARE_DEFAULT
#define HYDK_EMIT_SHARE_DEFAULT 1
#endif
constexpr int kEmitPer = 8;                  /* symbols per lane and batch */
constexpr int kEmitBatch = 64 * kEmitPer;    /* 512 */
constexpr int kEmitWin = kEmitBatch + 4;     /* 512 symbols x at most 32 bits = 512 words + alignment slack */

__global__ __launch_bounds__(kThreads) void k_rans_emit(const HydkLfJob *__restrict__ jobs, const uint32_t *sym_count_all,
                                                        const uint16_t *aux_all, const uint16_t *flags_all, uint32_t aux_pitch,
                                                        const uint32_t *final_state_all, const uint32_t *group_bits_all,
                                                        const uint64_t *offsets_all, uint8_t *payload, int preset_bits,
                                                        const uint32_t *status, int vblocks) {
    __shared__ uint32_t s_win[4][kEmitWin];
    HYDK_URGENT();
    /* a workgroup takes the virtual blocks blockIdx.x, blockIdx.x + gridDim.x, ... (four groups each, one per wavefront):
     * inside the pipelined loop a workgroup WAITS for its place — the transform kernels of fifteen other frames take
     * whatever a compute unit frees, and this kernel's queue gets its turn among theirs — far longer than it runs, so
     * fewer workgroups with more to do each finish sooner (launch_rans_emit, HYDAMD_EMIT_SHARE) */
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    auto one_block = [&](const int vb) {
    const int slot = vb >> 4;
    const int g = ((vb & 15) << 2) + wave;
    const int ngroups = jobs[slot].gcols * jobs[slot].grows;
    const size_t G = (size_t)slot * HYDK_GROUPS_PER_LFG + g;
    if (g >= ngroups || (*status & HYDK_STATUS_OVERFLOW))
        return;
    const void *tok = (const char *)jobs[slot].tokens + (size_t)g * jobs[slot].tok_cap * jobs[slot].rec_bytes;
    const uint4 *aux = (const uint4 *)(aux_all + G * aux_pitch);
    const uint16_t *flags = flags_all + G * (aux_pitch / 16);
    uint32_t *W = (uint32_t *)payload; /* 256-byte aligned allocation */
    uint32_t *win = s_win[wave];
    const int n = __builtin_amdgcn_readfirstlane((int)sym_count_all[G]);
    const uint32_t bits = group_bits_all[G];
    const unsigned long long lo_bit = offsets_all[G] * 8ull, hi_bit = lo_bit + (unsigned long long)((bits + 7u) & ~7u);
    /* words this section shares with its neighbours (ORed into; the scan kernel cleared them) */
    const uint32_t shared_lo = (lo_bit & 31ull) ? (uint32_t)(lo_bit >> 5) : 0xFFFFFFFFu;
    const uint32_t shared_hi = (hi_bit & 31ull) ? (uint32_t)((hi_bit - 1ull) >> 5) : 0xFFFFFFFFu;
    /* bit positions relative to the word that holds the section's first bit: they fit 32 bits */
    const uint32_t wbase = (uint32_t)(lo_bit >> 5);
    uint32_t cur = (uint32_t)(lo_bit & 31ull) + bits; /* stream start so far */
    uint32_t carry = 0;                               /* content of the partly filled word at cur >> 5 */

    bool bad = false;
    auto put = [&](uint32_t word, uint32_t v) {
        const uint32_t d = wbase + word;
        if (d == shared_lo || d == shared_hi)
            atomicOr(&W[d], v);
        else
            W[d] = v;
    };
    /* eight (value, bit count) per lane, in stream order by (lane, index); written in front of what is there */
    /* `before_stores` runs between the window's assembly and its stores: the place to take delivery of loads that were
     * requested earlier — vmcnt counts stores too, so a wait for loads placed BEHIND the stores waits for the stores' round
     * trip as well (this kernel runs one wavefront per SIMD: nothing hides it) */
    auto emit = [&](const uint32_t (&val)[kEmitPer], const uint32_t (&nb)[kEmitPer], auto &&before_stores) {
        uint32_t mine = 0;
#pragma unroll
        for (int j = 0; j < kEmitPer; j++)
            mine += nb[j];
        const uint32_t inc = scan64_inclusive(mine);
        const uint32_t total = __builtin_amdgcn_readlane(inc, 63);
        if (!total) {
            before_stores();
            return;
        }
        if (total > cur) { /* the records and refill flags hold more bits than the chain kernel counted for this section: the
                            * stages disagree about the group (a bug upstream).  Stop here instead of walking a window of
                            * 2^32 bits: the frame fails with an internal error */
            if (lane == 0)
                atomicOr((uint32_t *)status, HYDK_STATUS_INCONSISTENT);
            bad = true;
            return;
        }
        const uint32_t newcur = cur - total;
        const uint32_t wlo = newcur >> 5, whi = (cur - 1u) >> 5;
        const uint32_t nwords = whi - wlo + 1u;
        for (uint32_t i = lane; i < nwords; i += 64)
            win[i] = 0;
        __builtin_amdgcn_wave_barrier();
        if (lane == 0 && (cur & 31u))
            win[whi - wlo] = carry;
        /* win[] is private to this wave: lock-step execution + in-order LDS make the phases
         * visible to each other; the wave barriers only pin the compiler's ordering */
        __builtin_amdgcn_wave_barrier();
        uint32_t pos = newcur + (inc - mine) - wlo * 32u; /* window-relative position of the lane's first value */
#pragma unroll
        for (int j = 0; j < kEmitPer; j++) {
            if (nb[j]) {
                const uint32_t w = pos >> 5, sh = pos & 31u;
                const unsigned long long lo = (unsigned long long)val[j] << sh;
                if ((uint32_t)lo)
                    atomicOr(&win[w], (uint32_t)lo);
                if ((uint32_t)(lo >> 32))
                    atomicOr(&win[w + 1], (uint32_t)(lo >> 32));
            }
            pos += nb[j];
        }
        __builtin_amdgcn_wave_barrier();
        before_stores();
        const bool low_partial = (newcur & 31u) != 0;
        for (uint32_t i = lane + (low_partial ? 1u : 0u); i < nwords; i += 64)
            put(wlo + i, win[i]);
        carry = low_partial ? win[0] : 0u;
        __builtin_amdgcn_wave_barrier();
        cur = newcur;
    };

    /* the next batch's records, refill words and flags travel while this one's window is being assembled, and are taken
     * delivery of (`take`) before this one's stores */
    uint4 rec_n[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    uint4 a_n = {0, 0, 0, 0};
    uint32_t fl_n = 0;
    auto fetch = [&](int batch) {
        const int p0 = batch * kEmitBatch + kEmitPer * lane;
        if (batch >= 0 && p0 < n) { /* the arrays are padded to a multiple of 16 symbols: whole octets are readable */
            rec_n[0] = load_records4(tok, p0 >> 2);
            rec_n[1] = load_records4(tok, (p0 >> 2) + 1);
            a_n = aux[p0 >> 3];
            fl_n = flags[p0 >> 4];
        }
    };
    const int batches = (n + kEmitBatch - 1) / kEmitBatch;
    uint32_t rc[kEmitPer] = {0, 0, 0, 0, 0, 0, 0, 0}, ac[4] = {0, 0, 0, 0}, flc = 0; /* the batch in hand */
    auto take = [&] {
        const uint32_t r[kEmitPer] = {rec_n[0].x, rec_n[0].y, rec_n[0].z, rec_n[0].w, rec_n[1].x, rec_n[1].y, rec_n[1].z, rec_n[1].w};
        const uint32_t a[4] = {a_n.x, a_n.y, a_n.z, a_n.w};
#pragma unroll
        for (int j = 0; j < kEmitPer; j++) {
            rc[j] = r[j];
            asm volatile("" : "+v"(rc[j])); /* here, not where the compiler would first need it */
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            ac[j] = a[j];
            asm volatile("" : "+v"(ac[j]));
        }
        flc = fl_n;
        asm volatile("" : "+v"(flc));
    };
    fetch(batches - 1);
    take();
    for (int b = batches - 1; b >= 0; b--) {
        const int p0 = b * kEmitBatch + kEmitPer * lane;
        const uint32_t recs[kEmitPer] = {rc[0], rc[1], rc[2], rc[3], rc[4], rc[5], rc[6], rc[7]};
        const uint32_t aw[kEmitPer] = {ac[0] & 0xFFFFu, ac[0] >> 16, ac[1] & 0xFFFFu, ac[1] >> 16,
                                       ac[2] & 0xFFFFu, ac[2] >> 16, ac[3] & 0xFFFFu, ac[3] >> 16};
        const uint32_t flq = flc >> (p0 & 15);
        fetch(b - 1);
        uint32_t val[kEmitPer], nb[kEmitPer];
#pragma unroll
        for (int j = 0; j < kEmitPer; j++) {
            const bool valid = p0 + j < n;
            const bool refill = valid && ((flq >> j) & 1u);
            const uint32_t rbits = HYDK_REC32_RBITS(recs[j]), residue = recs[j] >> 16;
            /* refill p is written just before residue p (entropy.c:1134-1147): at most 16 + 16 bits for integer input */
            val[j] = refill ? (residue << 16) | aw[j] : residue;
            nb[j] = valid ? rbits + (refill ? 16u : 0u) : 0u;
        }
        emit(val, nb, take);
        if (bad) return;
    }
    {
        /* [preset id][final state, low half first] precede everything (encoder.c:945, entropy.c:1127-1130) */
        uint32_t val[kEmitPer] = {0, 0, 0, 0, 0, 0, 0, 0}, nb[kEmitPer] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (lane == 0) {
            val[0] = jobs[slot].preset;
            nb[0] = (uint32_t)preset_bits;
            if (n > 0) {
                val[1] = final_state_all[G];
                nb[1] = 32;
            }
        }
        emit(val, nb, [] {});
    }
    if (lane == 0 && (cur & 31u))
        put(cur >> 5, carry);
    }; /* one_block */
    for (int vb = blockIdx.x; vb < vblocks; vb += gridDim.x)
        one_block(vb);
}

/* ==========================================================================================
 * K3b: section sizes -> offsets (single block), then pack each section to its byte offset.
 * Sections are byte-padded with zeros, as hyd_bitwriter_flush does (bitwriter.c:144-150).
 * ======================================================================================== */
/* One workgroup of 256 threads: a larger one would have to wait for a compute unit with sixteen free
 * wave slots while other frames' transform workgroups keep taking whatever frees up (measured: this
 * kernel took 0.8 ms in the pipelined loop as a 1024-thread workgroup, against 15 us alone). */
/* ==========================================================================================
 * Frame bookkeeping that used to be runtime memsets and copies (each of those is a blit kernel of its
 * own in the stream).
 *   k_frame_begin: block 0 copies job descriptors from the pinned host ring into device memory; the
 *                  other blocks clear the frame's accumulator arena when the launch is the frame's first.
 *   k_publish:     the frame's totals and status, written by the device into pinned host memory.
 * ======================================================================================== */
__global__ __launch_bounds__(kThreads) void k_frame_begin(const uint32_t *__restrict__ host_jobs, uint32_t *__restrict__ d_jobs,
                                                          uint32_t job_words, uint4 *__restrict__ accum, uint32_t quads) {
    HYDK_URGENT();
    if (blockIdx.x == 0) {
        for (uint32_t i = threadIdx.x; i < job_words; i += kThreads)
            d_jobs[i] = host_jobs[i];
        return;
    }
    for (uint32_t i = (blockIdx.x - 1) * kThreads + threadIdx.x; i < quads; i += (gridDim.x - 1) * kThreads)
        accum[i] = make_uint4(0u, 0u, 0u, 0u);
}

__global__ __launch_bounds__(64) void k_publish(const uint64_t *total, uint64_t *h_total, const unsigned long long *lf_total,
                                                unsigned long long *h_lf_total, const uint32_t *status, uint32_t *h_status) {
    HYDK_URGENT();
    if (threadIdx.x == 0 && total)
        *h_total = *total;
    if (threadIdx.x == 1 && lf_total)
        *h_lf_total = *lf_total;
    if (threadIdx.x == 2 && status)
        *h_status = *status;
}

__global__ __launch_bounds__(kThreads) void k_scan_sections(const uint32_t *group_bits, int count, uint64_t *offsets,
                                                            uint64_t *total, uint8_t *payload, uint64_t payload_cap,
                                                            int clear_shared_words, uint32_t *status) {
    __shared__ uint64_t s_wave[4];
    HYDK_URGENT();
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int per = (count + kThreads - 1) / kThreads;
    const int lo = min(count, t * per), hi = min(count, lo + per);
    uint64_t sum = 0;
    for (int i = lo; i < hi; i++)
        sum += (group_bits[i] + 7u) >> 3;
    /* inclusive scan of the per-thread byte counts: inside a wave by shuffles, across the four waves through LDS */
    uint64_t inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint64_t v = __shfl_up(inc, d);
        if (lane >= d)
            inc += v;
    }
    if (lane == 63)
        s_wave[wave] = inc;
    __syncthreads();
    uint64_t before = 0, all = 0;
    for (int w = 0; w < 4; w++) {
        before += w < wave ? s_wave[w] : 0;
        all += s_wave[w];
    }
    const bool fits = all <= payload_cap && !(*status & HYDK_STATUS_OVERFLOW);
    uint64_t run = before + inc - sum;
    for (int i = lo; i < hi; i++) {
        const uint64_t nbytes = (group_bits[i] + 7u) >> 3;
        offsets[i] = run;
        if (clear_shared_words && fits && nbytes) {
            /* k_rans_emit ORs into the first / last word of a section when a neighbour owns part of it */
            if (run & 3u)
                ((uint32_t *)payload)[run >> 2] = 0;
            if ((run + nbytes) & 3u)
                ((uint32_t *)payload)[(run + nbytes - 1) >> 2] = 0;
        }
        run += nbytes;
    }
    if (t == kThreads - 1) {
        *total = fits ? all : 0;
        if (all > payload_cap)
            atomicOr(status, HYDK_STATUS_PAYLOAD); /* the host enlarges the payload and reruns the frame */
    }
}

__global__ __launch_bounds__(kThreads) void k_pack_sections(const uint32_t *bitbuf, uint32_t bit_pitch_words,
                                                            const uint32_t *group_bits, const uint64_t *offsets,
                                                            uint8_t *payload, const uint32_t *status) {
    const int G = blockIdx.x;
    const uint32_t bits = group_bits[G];
    if (!bits || (*status & HYDK_STATUS_OVERFLOW))
        return;
    const uint32_t nbytes = (bits + 7u) >> 3;
    const uint32_t *W = bitbuf + (size_t)G * bit_pitch_words;
    const uint32_t start = bit_pitch_words * 32u - bits;
    const uint32_t sw = start >> 5, sh = start & 31u;
    uint8_t *dst = payload + offsets[G];
    for (uint32_t k = threadIdx.x; k * 4u < nbytes; k += kThreads) {
        const uint32_t w0 = W[sw + k];
        const uint32_t w1 = sw + k + 1 < bit_pitch_words ? W[sw + k + 1] : 0u;
        const uint32_t v = sh ? (w0 >> sh) | (w1 << (32u - sh)) : w0;
        const uint32_t lim = min(4u, nbytes - k * 4u);
        for (uint32_t bidx = 0; bidx < lim; bidx++)
            dst[k * 4u + bidx] = (uint8_t)(v >> (8u * bidx));
    }
}

/* ==========================================================================================
 * Export: everything another process needs to put this context's LF groups into a frame, as one
 * self-describing blob in a caller-provided device buffer (include/hydrium_amd.h HydAmdBlobHeader /
 * HydAmdBlobSlot): what a multi-GPU job gathers, with one collective, to the rank that assembles.
 * grid = num_slots + 1 + kExportCopyBlocks, block = 256.
 * ======================================================================================== */
constexpr int kExportCopyBlocks = 120;
constexpr uint32_t kBlobMagic = 0x42445948u; /* "HYDB" */
constexpr int kBlobHeaderBytes = 64, kBlobSlotBytes = 16 + 36 + 12 + 256 + 4608 + (int)sizeof(HydkLfStream);

__global__ __launch_bounds__(kThreads) void k_export_frame(const HydkLfJob *__restrict__ jobs, const HydkTables *tabs,
                                                           const uint32_t *group_bits, const HydkLfStream *lf_streams,
                                                           const uint8_t *payload, const uint64_t *hf_total,
                                                           const uint8_t *lf_packed, const unsigned long long *lf_total,
                                                           const uint32_t *status, int num_slots, int lf_coded,
                                                           uint8_t *dst, uint64_t capacity, int view) {
    /* view: header and slot records only — the two byte strings stay where they are and the header carries their
     * addresses (for an assembler on the same device, behind this kernel in the same stream: hydamd_export_frame_owned) */
    const int t = threadIdx.x, b = blockIdx.x;
    const uint64_t hf_bytes = *hf_total, lf_bytes = lf_coded ? (uint64_t)*lf_total : 0;
    const uint64_t lf_off = (uint64_t)kBlobHeaderBytes + (uint64_t)num_slots * kBlobSlotBytes;
    const uint64_t hf_off = view ? lf_off : (lf_off + lf_bytes + 15ull) & ~15ull;
    const uint64_t total = view ? lf_off : hf_off + hf_bytes;
    const bool fits = view ? total <= capacity : total + 16 <= capacity; /* the copies below move whole 16-byte pieces */
    if (b < num_slots) {
        if (capacity < lf_off)
            return;
        uint32_t *rec = (uint32_t *)(dst + kBlobHeaderBytes + (size_t)b * kBlobSlotBytes);
        const HydkTables *tab = tabs + b;
        if (t == 0) {
            rec[0] = jobs[b].preset;
            rec[1] = tab->running_max_alphabet;
            rec[2] = tab->log_alphabet_size;
            rec[3] = tab->error;
        }
        if (t < HYDK_MAX_CLUSTERS)
            rec[4 + t] = tab->alphabet[t];
        if (t < 3)
            rec[13 + t] = 0;
        if (t < HYDK_GROUPS_PER_LFG)
            rec[16 + t] = group_bits[(size_t)b * HYDK_GROUPS_PER_LFG + t];
        for (int i = t; i < HYDK_MAX_CLUSTERS * HYDK_ALPHABET; i += kThreads)
            rec[80 + i] = (&tab->freq[0][0])[i];
        const uint32_t *lf = (const uint32_t *)(lf_streams + b);
        for (int i = t; i < (int)(sizeof(HydkLfStream) / 4); i += kThreads)
            rec[80 + HYDK_MAX_CLUSTERS * HYDK_ALPHABET + i] = lf_coded ? lf[i] : 0u;
        return;
    }
    if (b == num_slots) {
        if (t == 0 && capacity >= (uint64_t)kBlobHeaderBytes) {
            uint32_t *h32 = (uint32_t *)dst;
            uint64_t *h64 = (uint64_t *)dst;
            h32[0] = kBlobMagic;
            h32[1] = 1; /* layout version */
            h32[2] = (uint32_t)num_slots;
            h32[3] = *status | (fits ? 0u : HYDK_STATUS_PAYLOAD); /* an undersized blob reads like an undersized payload */
            h64[2] = hf_bytes;
            h64[3] = lf_bytes;
            h64[4] = total;
            h32[10] = view ? 0x101u : (uint32_t)lf_coded;
            for (int i = 11; i < 16; i++)
                h32[i] = 0;
            if (view) {
                h32[12] = (uint32_t)(uintptr_t)lf_packed;
                h32[13] = (uint32_t)((uintptr_t)lf_packed >> 32);
                h32[14] = (uint32_t)(uintptr_t)payload;
                h32[15] = (uint32_t)((uintptr_t)payload >> 32);
            }
        }
        return;
    }
    if (!fits || view)
        return;
    /* the two byte strings, 16 bytes per thread and step (both sources and both targets are 16-byte aligned) */
    const int cb = b - num_slots - 1;
    const size_t stride = (size_t)kExportCopyBlocks * kThreads;
    const uint4 *s1 = (const uint4 *)lf_packed;
    uint4 *d1 = (uint4 *)(dst + lf_off);
    for (size_t i = (size_t)cb * kThreads + t; i < (size_t)((lf_bytes + 15) >> 4); i += stride)
        d1[i] = s1[i];
    const uint4 *s2 = (const uint4 *)payload;
    uint4 *d2 = (uint4 *)(dst + hf_off);
    for (size_t i = (size_t)cb * kThreads + t; i < (size_t)((hf_bytes + 15) >> 4); i += stride)
        d2[i] = s2[i];
}

/* ==========================================================================================
 * Self-test: do the register evaluations of the format.c LUTs reproduce the host-built tables
 * bit for bit?  (If not, the launcher keeps the exact LUT-gather variant of K1.)
 * ======================================================================================== */
template <int XMODE>
__global__ void k_lut_selftest(const uint16_t *in_lut16, const float *bias_lut, int linear_light, uint32_t *mismatches) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 65536)
        return;
    uint32_t bad = 0;
    if (input_lut16_eval(i, linear_light) != in_lut16[i])
        bad++;
    if (((float)i * kUnit16 <= 0.0404482362771082f) != (i <= kDarkMax))
        bad++;
    if (linear_light ? input_lut16_eval_as<kCurveNone>(i) != in_lut16[i]
                     : (i > kDarkMax ? input_lut16_eval_as<kCurveCubic>(i) : input_lut16_eval_as<kCurveBoth>(i)) != in_lut16[i])
        bad++;
    if (__float_as_uint(bias_lut_eval<XMODE>(i)) != __float_as_uint(bias_lut[i]))
        bad++;
    if (bad)
        atomicAdd(mismatches, bad);
}

/* ------------------------------------------------------------------------------------------
 * launch helpers (C++ linkage, used by device_api.hip)
 * ---------------------------------------------------------------------------------------- */
namespace hydk {

/* One launch per template instance that owns at least one LF group of this round; an instance
 * returns at once for the LF groups of another sample format. */
hipError_t launch_transform(const HydkLfJob *d_jobs, int num_slots, unsigned fmt_mask, int xmode, uint32_t *status,
                            uint2 *part_info, int plog, hipStream_t stream) {
    const dim3 grid((num_slots * HYDK_GROUPS_PER_LFG) << plog), block(kThreads);
#define HYDK_LAUNCH_K1(FMT, XM) hipLaunchKernelGGL((k_transform_tokenize<FMT, XM>), grid, block, 0, stream, d_jobs, status, part_info, plog)
#define HYDK_LAUNCH_K1_MODES(FMT)              \
    do {                                       \
        if (xmode == kXybFastRcp)              \
            HYDK_LAUNCH_K1(FMT, kXybFastRcp);  \
        else if (xmode == kXybIeeeDiv)         \
            HYDK_LAUNCH_K1(FMT, kXybIeeeDiv);  \
        else if (xmode == kXybFastRcpRegs)     \
            HYDK_LAUNCH_K1(FMT, kXybFastRcpRegs); \
        else if (xmode == kXybIeeeDivRegs)     \
            HYDK_LAUNCH_K1(FMT, kXybIeeeDivRegs); \
        else                                   \
            HYDK_LAUNCH_K1(FMT, kXybGather);   \
    } while (0)
    if (fmt_mask & (1u << HYDK_FMT_U8))
        HYDK_LAUNCH_K1_MODES(HYDK_FMT_U8);
    if (fmt_mask & (1u << HYDK_FMT_U16))
        HYDK_LAUNCH_K1_MODES(HYDK_FMT_U16);
    if (fmt_mask & (1u << HYDK_FMT_F32))
        HYDK_LAUNCH_K1(HYDK_FMT_F32, kXybIeeeDiv);
#undef HYDK_LAUNCH_K1_MODES
#undef HYDK_LAUNCH_K1
    if (plog)
        hipLaunchKernelGGL(k_join_parts, dim3(num_slots * HYDK_GROUPS_PER_LFG), block, 0, stream, d_jobs, part_info, plog, status);
    return hipGetLastError();
}

hipError_t transform_footprint(int fmt, int xmode, int *lds_bytes, int *registers) {
    const void *fn = nullptr;
#define HYDK_K1_PTR(FMT, XM) fn = (const void *)k_transform_tokenize<FMT, XM>
    if (fmt == HYDK_FMT_F32)
        HYDK_K1_PTR(HYDK_FMT_F32, kXybIeeeDiv);
    else if (fmt == HYDK_FMT_U8) {
        if (xmode == kXybFastRcp)
            HYDK_K1_PTR(HYDK_FMT_U8, kXybFastRcp);
        else if (xmode == kXybIeeeDiv)
            HYDK_K1_PTR(HYDK_FMT_U8, kXybIeeeDiv);
        else
            HYDK_K1_PTR(HYDK_FMT_U8, kXybGather);
    } else {
        if (xmode == kXybFastRcp)
            HYDK_K1_PTR(HYDK_FMT_U16, kXybFastRcp);
        else if (xmode == kXybIeeeDiv)
            HYDK_K1_PTR(HYDK_FMT_U16, kXybIeeeDiv);
        else
            HYDK_K1_PTR(HYDK_FMT_U16, kXybGather);
    }
#undef HYDK_K1_PTR
    hipFuncAttributes attr;
    const hipError_t e = hipFuncGetAttributes(&attr, fn);
    if (e == hipSuccess) {
        *lds_bytes = (int)attr.sharedSizeBytes;
        *registers = attr.numRegs;
    }
    return e;
}

/* lf_hist != NULL: the launch also builds the LF coder's prefix codes of the same LF groups (num_slots more workgroups);
 * lf_hist / lf_streams / lf_work already point at the first of them */
hipError_t launch_tables(const uint32_t *hist, HydkTables *tabs, const uint32_t *alpha_max, int nclusters, int first_slot,
                         int num_slots, uint32_t alpha_floor, const uint32_t *alpha_floor_dev, const uint32_t *lf_hist,
                         HydkLfStream *lf_streams, void *lf_work, int slots_per_frame, hipStream_t stream) {
    hipLaunchKernelGGL(k_build_tables, dim3(lf_hist ? 2 * num_slots : num_slots), dim3(kThreads), 0, stream, hist, tabs, alpha_max,
                       nclusters, alpha_floor, alpha_floor_dev, first_slot, num_slots, lf_hist, lf_streams, lf_work, slots_per_frame);
    return hipGetLastError();
}

hipError_t launch_rans(const HydkLfJob *d_jobs, const uint32_t *sym_count, const HydkTables *tabs, uint32_t *bitbuf,
                       uint32_t bit_pitch_words, uint32_t *group_bits, int preset_bits, int num_slots, const uint32_t *status,
                       hipStream_t stream) {
    hipLaunchKernelGGL((k_rans_encode<4, false>), dim3(num_slots * 16), dim3(256), 0, stream, d_jobs, sym_count, tabs, bitbuf,
                       bit_pitch_words, group_bits, preset_bits, status, (uint16_t *)nullptr, (uint16_t *)nullptr, 0u,
                       (uint32_t *)nullptr);
    return hipGetLastError();
}

/* the same walk leaving refill words and flags for k_rans_emit instead of writing bits itself (4-byte records only) */
hipError_t launch_rans_deferred(const HydkLfJob *d_jobs, const uint32_t *sym_count, const HydkTables *tabs, uint16_t *aux,
                                uint16_t *flags, uint32_t aux_pitch, uint32_t *final_state, uint32_t *group_bits,
                                int preset_bits, int num_slots, const uint32_t *status, hipStream_t stream) {
    hipLaunchKernelGGL((k_rans_encode<4, true>), dim3(num_slots * 16), dim3(256), 0, stream, d_jobs, sym_count, tabs,
                       (uint32_t *)nullptr, 0u, group_bits, preset_bits, status, aux, flags, aux_pitch, final_state);
    return hipGetLastError();
}

/* lf_hist != NULL: the launch also builds the LF coder's prefix codes of the same LF groups (num_slots more workgroups) */
hipError_t launch_rans_lanes(const HydkLfJob *d_jobs, const uint32_t *sym_count, const HydkTables *tabs, uint16_t *aux,
                             uint16_t *flags, uint32_t aux_pitch, uint32_t *final_state, uint32_t *group_bits, int preset_bits,
                             int nclusters, int num_slots, const uint32_t *status, const uint32_t *lf_hist,
                             HydkLfStream *lf_streams, void *lf_work, hipStream_t stream) {
    const dim3 grid(lf_hist ? 2 * num_slots : num_slots);
#if HYDK_CHAIN_DYN_LDS
#define HYDK_LAUNCH_LANES(NC)                                                                                                  \
    do {                                                                                                                       \
        /* more than 64 KB of dynamic LDS is opted into per function AND device (several devices in one process: multi.c) */ \
        static std::atomic<uint32_t> opted{0};                                                                                 \
        int dev = 0;                                                                                                           \
        if (lanes_lds_bytes(NC) > 65536 && hipGetDevice(&dev) == hipSuccess && !((opted.load() >> (dev & 31)) & 1u)) {         \
            const hipError_t e = hipFuncSetAttribute((const void *)k_rans_lanes<NC>, hipFuncAttributeMaxDynamicSharedMemorySize, lanes_lds_bytes(NC)); \
            if (e != hipSuccess)                                                                                               \
                return e;                                                                                                      \
            opted.fetch_or(1u << (dev & 31));                                                                                  \
        }                                                                                                                      \
        hipLaunchKernelGGL(k_rans_lanes<NC>, grid, dim3(64), (size_t)lanes_lds_bytes(NC), stream, d_jobs, sym_count, tabs, aux, flags, aux_pitch, \
                           final_state, group_bits, preset_bits, status, num_slots, lf_hist, lf_streams, lf_work);             \
    } while (0)
#else
#define HYDK_LAUNCH_LANES(NC)                                                                                                  \
    hipLaunchKernelGGL(k_rans_lanes<NC>, grid, dim3(64), 0, stream, d_jobs, sym_count, tabs, aux, flags, aux_pitch, final_state, \
                       group_bits, preset_bits, status, num_slots, lf_hist, lf_streams, lf_work)
#endif
    /* the tables in LDS are sized by the clustering scheme (encoder.c:862-901: 9 / 3 / 2 / 1 clusters per preset) */
    if (nclusters == 9)
        HYDK_LAUNCH_LANES(HYDK_LANE_NC9_PROBE);
    else if (nclusters == 3)
        HYDK_LAUNCH_LANES(3);
    else if (nclusters == 2)
        HYDK_LAUNCH_LANES(2);
    else if (nclusters == 1)
        HYDK_LAUNCH_LANES(1);
    else
        return hipErrorInvalidValue;
#undef HYDK_LAUNCH_LANES
    return hipGetLastError();
}

hipError_t launch_rans_emit(const HydkLfJob *d_jobs, const uint32_t *sym_count, const uint16_t *aux, const uint16_t *flags,
                            uint32_t aux_pitch, const uint32_t *final_state, const uint32_t *group_bits, const uint64_t *offsets,
                            uint8_t *payload, int preset_bits, int num_slots, const uint32_t *status, hipStream_t stream) {
    /* virtual blocks per workgroup: HYDAMD_EMIT_SHARE (A/B; 1 = one workgroup per four groups, as until round 5) */
    static const int share = [] {
        const char *v = getenv("HYDAMD_EMIT_SHARE");
        const int n = v && *v ? atoi(v) : HYDK_EMIT_SHARE_DEFAULT;
        return n < 1 ? 1 : n > 16 ? 16 : n;
    }();
    const int vblocks = num_slots * 16;
    hipLaunchKernelGGL(k_rans_emit, dim3((vblocks + share - 1) / share), dim3(kThreads), 0, stream, d_jobs, sym_count, aux, flags,
                       aux_pitch, final_state, group_bits, offsets, payload, preset_bits, status, vblocks);
    return hipGetLastError();
}

hipError_t launch_frame_begin(const HydkLfJob *host_jobs, HydkLfJob *d_jobs, int count, uint32_t *accum, size_t accum_words,
                              hipStream_t stream) {
    const uint32_t quads = accum ? (uint32_t)(accum_words / 4) : 0u;
    const uint32_t job_words = (uint32_t)((size_t)count * sizeof(HydkLfJob) / sizeof(uint32_t));
    const uint32_t blocks = 1u + (quads + kThreads - 1) / kThreads;
    hipLaunchKernelGGL(k_frame_begin, dim3(blocks < 64u ? blocks : 64u), dim3(kThreads), 0, stream, (const uint32_t *)host_jobs,
                       (uint32_t *)d_jobs, job_words, (uint4 *)accum, quads);
    return hipGetLastError();
}

hipError_t launch_publish(const uint64_t *total, uint64_t *h_total, const unsigned long long *lf_total,
                          unsigned long long *h_lf_total, const uint32_t *status, uint32_t *h_status, hipStream_t stream) {
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, stream, total, h_total, lf_total, h_lf_total, status, h_status);
    return hipGetLastError();
}

hipError_t launch_scan(const uint32_t *group_bits, int count, uint64_t *offsets, uint64_t *total, uint8_t *payload,
                       uint64_t payload_cap, int clear_shared_words, uint32_t *status, hipStream_t stream) {
    hipLaunchKernelGGL(k_scan_sections, dim3(1), dim3(kThreads), 0, stream, group_bits, count, offsets, total, payload, payload_cap,
                       clear_shared_words, status);
    return hipGetLastError();
}

hipError_t launch_pack(const uint32_t *bitbuf, uint32_t bit_pitch_words, const uint32_t *group_bits, const uint64_t *offsets,
                       uint8_t *payload, int count, const uint32_t *status, hipStream_t stream) {
    hipLaunchKernelGGL(k_pack_sections, dim3(count), dim3(kThreads), 0, stream, bitbuf, bit_pitch_words, group_bits, offsets,
                       payload, status);
    return hipGetLastError();
}

hipError_t launch_export(const HydkLfJob *d_jobs, const HydkTables *tabs, const uint32_t *group_bits, const HydkLfStream *lf_streams,
                         const uint8_t *payload, const uint64_t *hf_total, const uint8_t *lf_packed,
                         const unsigned long long *lf_total, const uint32_t *status, int num_slots, int lf_coded, uint8_t *dst,
                         uint64_t capacity, int view, hipStream_t stream) {
    hipLaunchKernelGGL(k_export_frame, dim3(num_slots + 1 + (view ? 0 : kExportCopyBlocks)), dim3(kThreads), 0, stream, d_jobs, tabs,
                       group_bits, lf_streams, payload, hf_total, lf_packed, lf_total, status, num_slots, lf_coded, dst, capacity, view);
    return hipGetLastError();
}

hipError_t launch_lut_selftest(const uint16_t *in_lut16, const float *bias_lut, int linear_light, int xmode,
                               uint32_t *mismatches, hipStream_t stream) {
    if (xmode == kXybFastRcp)
        hipLaunchKernelGGL(k_lut_selftest<kXybFastRcp>, dim3(256), dim3(256), 0, stream, in_lut16, bias_lut, linear_light,
                           mismatches);
    else
        hipLaunchKernelGGL(k_lut_selftest<kXybIeeeDiv>, dim3(256), dim3(256), 0, stream, in_lut16, bias_lut, linear_light,
                           mismatches);
    return hipGetLastError();
}

} /* namespace hydk */
