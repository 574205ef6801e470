// VALU issue-rate and dependent-latency table for gfx950, one instruction kind at a time, written
// in inline asm so that -ffp-contract / SLP decisions of the compiler cannot change what is measured.
//
//   throughput: 8 independent register chains per wave, 8 waves per SIMD (2048 threads per CU),
//               every CU busy -> lane-ops/s and cycles per wave64 instruction per SIMD
//   latency:    ONE wave on the chip, one dependent chain -> s_memtime ticks per instruction
//
// Build: hipcc --offload-arch=gfx950 -O2 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

typedef float float2_ __attribute__((ext_vector_type(2)));

// ---- instruction bodies: R(i) names accumulator i; each macro issues ONE instruction on it ----
// scalar f32
#define I_MUL(a)      asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(k0))
#define I_ADD(a)      asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(k1))
#define I_FMA(a)      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(k0), "v"(k1))
// packed f32 (a is a float2_)
#define I_PKMUL(a)    asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a) : "v"(p0))
#define I_PKADD(a)    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(p1))
#define I_PKFMA(a)    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(p0), "v"(p1))
// integer
#define I_MAD24(a)    asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a) : "v"(u0), "v"(u1))
#define I_MULLO(a)    asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(u0))
#define I_MULHI(a)    asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a) : "v"(u0))
#define I_MULHI24(a)  asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a) : "v"(u0))
#define I_ADDU(a)     asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(u0))
#define I_ADD3(a)     asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a) : "v"(u0), "v"(u1))
#define I_LSHLADD(a)  asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a) : "v"(u0))
#define I_BFE(a)      asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(a))
#define I_PERM(a)     asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a) : "v"(u0), "v"(u1))
#define I_CNDMASK(a)  asm volatile("v_cmp_gt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(u0) : "vcc") /* two instructions */
#define I_CMP(a)      asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(a), "v"(u0) : "vcc")
#define I_LSHL64(a)   asm volatile("v_lshlrev_b64 %0, %1, %0" : "+v"(a) : "v"(u0))
#define I_BCNT(a)     asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a) : "v"(u0))
#define I_FFBH(a)     asm volatile("v_ffbh_u32 %0, %0" : "+v"(a))
// conversions / transcendental
#define I_CVTFU(a)    asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a))
#define I_CVTIF(a)    asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a))
#define I_RCP(a)      asm volatile("v_rcp_f32 %0, %0" : "+v"(a))
#define I_MED3(a)     asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(a) : "v"(u0), "v"(u1))
// cross-lane
#define I_DPPSHR(a)   asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a))
#define I_DPPADD(a)   asm volatile("v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a))
#define I_SDWA(a)     asm volatile("v_cvt_f32_u32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(a))

#define I_AND(a) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a) : "v"(u0))
#define I_OR(a) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a) : "v"(u1))
#define I_XOR(a) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a) : "v"(u0))
#define I_LSHL(a) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a))
#define I_LSHR(a) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(a))
#define I_ASHR(a) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(a))
#define I_SUBU(a) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a) : "v"(u1))
#define I_MINU(a) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a) : "v"(u0))
#define I_MAXF(a) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a) : "v"(k1))
#define I_SUBF(a) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a) : "v"(k1))
#define I_FMAC(a) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a) : "v"(k0), "v"(k1))
#define I_CNDONLY(a) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(u0) : "vcc")
#define I_MOV(a) asm volatile("v_mov_b32 %0, %1" : "+v"(a) : "v"(u0))
#define I_ANDOR(a) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a) : "v"(u0), "v"(u1))
#define I_OR3(a) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a) : "v"(u0), "v"(u1))
#define I_LSHLOR(a) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(a) : "v"(u1))
#define I_CMPF(a) asm volatile("v_cmp_ge_f32 vcc, %0, %1" : : "v"(a), "v"(k1) : "vcc")
#define I_CMPFABS(a) asm volatile("v_cmp_ge_f32_e64 vcc, |%0|, %1" : : "v"(a), "v"(k1) : "vcc")
#define I_MUL24(a) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a) : "v"(u1))
#define I_TRUNC(a) asm volatile("v_trunc_f32 %0, %0" : "+v"(a))
#define I_FLOOR(a) asm volatile("v_floor_f32 %0, %0" : "+v"(a))
#define I_CVTUF(a) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(a))
#define I_ADDC(a) asm volatile("v_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(a) : : "vcc")
#define I_XAD(a) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(a) : "v"(u0), "v"(u1))
#define I_BFI(a) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a) : "v"(u0), "v"(u1))
#define I_ALIGNBIT(a) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a) : "v"(u0))
#define I_MULABS(a) asm volatile("v_mul_f32_e64 %0, |%0|, %1" : "+v"(a) : "v"(k0))
#define I_MUL_LIT(a) asm volatile("v_mul_f32 %0, 0x3e318a87, %0" : "+v"(a))
#define I_ADD_SGPR(a) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a) : "s"(k1))

#define REP8(M, T) M(T[0]); M(T[1]); M(T[2]); M(T[3]); M(T[4]); M(T[5]); M(T[6]); M(T[7])
#define REP8x2(M1, M2, T) M1(T[0]); M1(T[1]); M1(T[2]); M1(T[3]); M1(T[4]); M1(T[5]); M1(T[6]); M1(T[7]); \
                          M2(T[0]); M2(T[1]); M2(T[2]); M2(T[3]); M2(T[4]); M2(T[5]); M2(T[6]); M2(T[7])

enum Kind { K_MUL_ADD, K_FMA, K_PKMUL_PKADD, K_PKFMA, K_MAD24, K_MULLO, K_MULHI, K_MULHI24, K_ADDU, K_ADD3, K_LSHLADD,
            K_BFE, K_PERM, K_CNDMASK, K_CMP, K_LSHL64, K_BCNT, K_FFBH, K_CVTFU, K_CVTIF, K_RCP, K_MED3, K_DPPSHR,
            K_DPPADD, K_SDWA, K_MUL_ONLY, K_ADD_ONLY, K_PKMUL_ONLY, K_AND, K_OR, K_XOR, K_LSHL, K_LSHR, K_ASHR, K_SUBU, K_MINU, K_MAXF, K_SUBF, K_FMAC, K_CNDONLY, K_MOV, K_ANDOR, K_OR3, K_LSHLOR, K_CMPF, K_CMPFABS, K_MUL24, K_TRUNC, K_FLOOR, K_CVTUF, K_ADDC, K_XAD, K_BFI, K_ALIGNBIT, K_MULABS, K_MUL_LIT, K_ADD_SGPR, K_COUNT };
static const char *kNames[K_COUNT] = {
    "v_mul_f32 + v_add_f32 (non-fused pair)", "v_fma_f32", "v_pk_mul_f32 + v_pk_add_f32", "v_pk_fma_f32", "v_mad_u32_u24",
    "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_u32_u24", "v_add_u32", "v_add3_u32", "v_lshl_add_u32", "v_bfe_u32", "v_perm_b32",
    "v_cmp_gt_u32 + v_cndmask_b32 (PAIR: halve)", "v_cmp_gt_u32 (vcc)", "v_lshlrev_b64", "v_bcnt_u32_b32", "v_ffbh_u32", "v_cvt_f32_u32", "v_cvt_i32_f32",
    "v_rcp_f32", "v_med3_i32", "v_mov_b32_dpp row_shr:1", "v_add_u32_dpp row_shr:1", "v_cvt_f32_u32_sdwa WORD_1",
    "v_mul_f32", "v_add_f32", "v_pk_mul_f32",
    "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_sub_u32", "v_min_u32", "v_max_f32", "v_sub_f32", "v_fmac_f32", "v_cndmask_b32 (vcc untouched)", "v_mov_b32", "v_and_or_b32", "v_or3_b32", "v_lshl_or_b32", "v_cmp_ge_f32 (vcc)", "v_cmp_ge_f32 |x| (VOP3, vcc)", "v_mul_u32_u24", "v_trunc_f32", "v_floor_f32", "v_cvt_u32_f32", "v_addc_co_u32 (vcc in/out)", "v_xad_u32", "v_bfi_b32", "v_alignbit_b32", "v_mul_f32 |x| (VOP3)", "v_mul_f32 with 32-bit literal", "v_add_f32 with SGPR operand"};
// lane-level arithmetic results per instruction (2 for packed forms)
static const int kLaneOps[K_COUNT] = {1, 1, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};

template <int KIND, int CHAINS /* 8: throughput, 1: dependent latency */>
__global__ __launch_bounds__(256) void k(float *out, long long *cyc, int iters, float seed) {
    const float k0 = 1.0001f, k1 = 0.25f;
    const float2_ p0 = {1.0001f, 0.9999f}, p1 = {0.25f, 0.125f};
    const unsigned u0 = 0x10003u, u1 = 7u;
    float f[8];
    float2_ pf[8];
    unsigned u[8];
    unsigned long long w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        f[i] = seed + threadIdx.x * 1e-3f + i;
        pf[i] = float2_{f[i], f[i] + 0.5f};
        u[i] = threadIdx.x * 2654435761u + i;
        w[i] = u[i];
    }
    long long w0 = wall_clock64();
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#define BODY8(M, T) if (CHAINS == 8) { REP8(M, T); REP8(M, T); } else { M(T[0]); M(T[0]); M(T[0]); M(T[0]); M(T[0]); M(T[0]); M(T[0]); M(T[0]); M(T[0]); M(T[0]); M(T[0]); M(T[0]); M(T[0]); M(T[0]); M(T[0]); M(T[0]); }
#define BODY8x2(M1, M2, T) if (CHAINS == 8) { REP8x2(M1, M2, T); } else { M1(T[0]); M2(T[0]); M1(T[0]); M2(T[0]); M1(T[0]); M2(T[0]); M1(T[0]); M2(T[0]); M1(T[0]); M2(T[0]); M1(T[0]); M2(T[0]); M1(T[0]); M2(T[0]); M1(T[0]); M2(T[0]); }
        // every body issues exactly 16 instructions per iteration
        if (KIND == K_MUL_ADD) { BODY8x2(I_MUL, I_ADD, f) }
        else if (KIND == K_FMA) { BODY8(I_FMA, f) }
        else if (KIND == K_PKMUL_PKADD) { BODY8x2(I_PKMUL, I_PKADD, pf) }
        else if (KIND == K_PKFMA) { BODY8(I_PKFMA, pf) }
        else if (KIND == K_MAD24) { BODY8(I_MAD24, u) }
        else if (KIND == K_MULLO) { BODY8(I_MULLO, u) }
        else if (KIND == K_MULHI) { BODY8(I_MULHI, u) }
        else if (KIND == K_MULHI24) { BODY8(I_MULHI24, u) }
        else if (KIND == K_ADDU) { BODY8(I_ADDU, u) }
        else if (KIND == K_ADD3) { BODY8(I_ADD3, u) }
        else if (KIND == K_LSHLADD) { BODY8(I_LSHLADD, u) }
        else if (KIND == K_BFE) { BODY8(I_BFE, u) }
        else if (KIND == K_PERM) { BODY8(I_PERM, u) }
        else if (KIND == K_CNDMASK) { BODY8(I_CNDMASK, u) }
        else if (KIND == K_CMP) { BODY8(I_CMP, u) }
        else if (KIND == K_LSHL64) { BODY8(I_LSHL64, w) }
        else if (KIND == K_BCNT) { BODY8(I_BCNT, u) }
        else if (KIND == K_FFBH) { BODY8(I_FFBH, u) }
        else if (KIND == K_CVTFU) { BODY8(I_CVTFU, u) }
        else if (KIND == K_CVTIF) { BODY8(I_CVTIF, u) }
        else if (KIND == K_RCP) { BODY8(I_RCP, f) }
        else if (KIND == K_MED3) { BODY8(I_MED3, u) }
        else if (KIND == K_DPPSHR) { BODY8(I_DPPSHR, u) }
        else if (KIND == K_DPPADD) { BODY8(I_DPPADD, u) }
        else if (KIND == K_SDWA) { BODY8(I_SDWA, u) }
        else if (KIND == K_MUL_ONLY) { BODY8(I_MUL, f) }
        else if (KIND == K_ADD_ONLY) { BODY8(I_ADD, f) }
        else if (KIND == K_PKMUL_ONLY) { BODY8(I_PKMUL, pf) }
        else if (KIND == K_AND) { BODY8(I_AND, u) }
        else if (KIND == K_OR) { BODY8(I_OR, u) }
        else if (KIND == K_XOR) { BODY8(I_XOR, u) }
        else if (KIND == K_LSHL) { BODY8(I_LSHL, u) }
        else if (KIND == K_LSHR) { BODY8(I_LSHR, u) }
        else if (KIND == K_ASHR) { BODY8(I_ASHR, u) }
        else if (KIND == K_SUBU) { BODY8(I_SUBU, u) }
        else if (KIND == K_MINU) { BODY8(I_MINU, u) }
        else if (KIND == K_MAXF) { BODY8(I_MAXF, f) }
        else if (KIND == K_SUBF) { BODY8(I_SUBF, f) }
        else if (KIND == K_FMAC) { BODY8(I_FMAC, f) }
        else if (KIND == K_CNDONLY) { BODY8(I_CNDONLY, u) }
        else if (KIND == K_MOV) { BODY8(I_MOV, u) }
        else if (KIND == K_ANDOR) { BODY8(I_ANDOR, u) }
        else if (KIND == K_OR3) { BODY8(I_OR3, u) }
        else if (KIND == K_LSHLOR) { BODY8(I_LSHLOR, u) }
        else if (KIND == K_CMPF) { BODY8(I_CMPF, f) }
        else if (KIND == K_CMPFABS) { BODY8(I_CMPFABS, f) }
        else if (KIND == K_MUL24) { BODY8(I_MUL24, u) }
        else if (KIND == K_TRUNC) { BODY8(I_TRUNC, f) }
        else if (KIND == K_FLOOR) { BODY8(I_FLOOR, f) }
        else if (KIND == K_CVTUF) { BODY8(I_CVTUF, u) }
        else if (KIND == K_ADDC) { BODY8(I_ADDC, u) }
        else if (KIND == K_XAD) { BODY8(I_XAD, u) }
        else if (KIND == K_BFI) { BODY8(I_BFI, u) }
        else if (KIND == K_ALIGNBIT) { BODY8(I_ALIGNBIT, u) }
        else if (KIND == K_MULABS) { BODY8(I_MULABS, f) }
        else if (KIND == K_MUL_LIT) { BODY8(I_MUL_LIT, f) }
        else if (KIND == K_ADD_SGPR) { BODY8(I_ADD_SGPR, f) }
    }
    long long t1 = __builtin_readcyclecounter();
    long long w1 = wall_clock64();
    float acc = 0;
#pragma unroll
    for (int i = 0; i < 8; i++)
        acc += f[i] + pf[i].x + pf[i].y + (float)u[i] + (float)(unsigned)w[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        cyc[0] = t1 - t0;
        cyc[1] = w1 - w0; /* constant-rate counter (hipDeviceAttributeWallClockRate) */
    }
}

static bool g_swept = false;

template <int KIND>
static void run_kind(float *out, long long *cyc, int cus, double wall_khz) {
    const int iters_tp = 4000, iters_lat = 20000;
    if (!g_swept) { /* once: how does the issue rate grow with the waves sharing a SIMD? */
        g_swept = true;
        for (int per_cu = 1; per_cu <= 8; per_cu *= 2) {
            hipEvent_t a, b;
            CK(hipEventCreate(&a));
            CK(hipEventCreate(&b));
            hipLaunchKernelGGL((k<KIND, 8>), dim3(cus * per_cu), dim3(256), 0, 0, out, cyc, iters_tp, 1.0f);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(a));
            hipLaunchKernelGGL((k<KIND, 8>), dim3(cus * per_cu), dim3(256), 0, 0, out, cyc, iters_tp, 1.0f);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            long long c2[2];
            CK(hipMemcpy(c2, cyc, 16, hipMemcpyDeviceToHost));
            printf("  [%s] %d waves/SIMD: kernel %.3f ms, wave 0: %lld s_memtime ticks = %lld wall-clock ticks (100 MHz) -> s_memtime runs at %.1f MHz; %.2f ns per wave-instr per SIMD\n",
                   kNames[KIND], per_cu, ms, c2[0], c2[1], c2[1] ? (double)c2[0] / c2[1] * 100.0 : 0.0,
                   ms * 1e6 / ((double)per_cu * iters_tp * 16));
        }
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // throughput: 8 blocks of 256 threads per CU = 8 waves per SIMD
    const int blocks = cus * 8;
    hipLaunchKernelGGL((k<KIND, 8>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters_tp, 1.0f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<KIND, 8>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters_tp, 1.0f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    long long ticks;
    CK(hipMemcpy(&ticks, cyc, 8, hipMemcpyDeviceToHost));
    const double wave_instr = (double)blocks * 4 * iters_tp * 16;       // wave-instructions issued
    const double per_simd = wave_instr / (cus * 4.0);                   // per SIMD
    const double ns_per_instr = ms * 1e6 / per_simd;
    const double lane_ops = wave_instr * 64 * kLaneOps[KIND];
    // latency: one wave, one chain
    hipLaunchKernelGGL((k<KIND, 1>), dim3(1), dim3(64), 0, 0, out, cyc, iters_lat, 1.0f);
    CK(hipDeviceSynchronize());
    long long lt;
    CK(hipMemcpy(&lt, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-40s | %7.2f T lane-ops/s | %6.3f ns per wave-instr per SIMD | kernel %8.3f ms, %6.2f ticks/instr in-kernel | dependent: %6.2f ticks\n",
           kNames[KIND], lane_ops / (ms * 1e-3) / 1e12, ns_per_instr, ms, (double)ticks / (iters_tp * 16.0) / 8.0 /* 8 waves share a SIMD */,
           (double)lt / (iters_lat * 16.0));
    (void)wall_khz;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    int clk = 0, wclk = 0;
    CK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0));
    CK(hipDeviceGetAttribute(&wclk, hipDeviceAttributeWallClockRate, 0));
    printf("%s: %d CUs, shader clock %d kHz, s_memtime clock %d kHz\n", prop.gcnArchName, cus, clk, wclk);
    printf("ticks are s_memtime ticks (%.1f MHz): multiply by shader clock / that to get shader cycles\n", wclk / 1e3);
    float *out;
    long long *cyc;
    CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    CK(hipMalloc(&cyc, 16));
#define RUN(K) run_kind<K>(out, cyc, cus, wclk);
    RUN(K_MUL_ADD) RUN(K_MUL_ONLY) RUN(K_ADD_ONLY) RUN(K_FMA) RUN(K_PKMUL_PKADD) RUN(K_PKMUL_ONLY) RUN(K_PKFMA)
    RUN(K_MAD24) RUN(K_MULLO) RUN(K_MULHI) RUN(K_MULHI24) RUN(K_ADDU) RUN(K_ADD3) RUN(K_LSHLADD) RUN(K_BFE) RUN(K_PERM)
    RUN(K_CNDMASK) RUN(K_CMP) RUN(K_LSHL64) RUN(K_BCNT) RUN(K_FFBH) RUN(K_CVTFU) RUN(K_CVTIF) RUN(K_RCP) RUN(K_MED3)
    RUN(K_DPPSHR) RUN(K_DPPADD) RUN(K_SDWA)
    RUN(K_AND) RUN(K_OR) RUN(K_XOR) RUN(K_LSHL) RUN(K_LSHR) RUN(K_ASHR) RUN(K_SUBU) RUN(K_MINU) RUN(K_MAXF) RUN(K_SUBF) RUN(K_FMAC) RUN(K_CNDONLY) RUN(K_MOV) RUN(K_ANDOR) RUN(K_OR3) RUN(K_LSHLOR) RUN(K_CMPF) RUN(K_CMPFABS) RUN(K_MUL24) RUN(K_TRUNC) RUN(K_FLOOR) RUN(K_CVTUF) RUN(K_ADDC) RUN(K_XAD) RUN(K_BFI) RUN(K_ALIGNBIT) RUN(K_MULABS) RUN(K_MUL_LIT) RUN(K_ADD_SGPR)
    return 0;
}
