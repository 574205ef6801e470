// Latency anatomy of the rANS recurrence on one wave: full step, step without the LDS lookup,
// bare dependent LDS read chain, bare dependent VALU chain.  Reports cycles per step (s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); exit(1);} } while (0)

template <int MODE>
__global__ __launch_bounds__(64) void k(unsigned *out, long long *cyc, int iters, unsigned f, unsigned mg, int lanes = 64) {
    __shared__ unsigned short tab[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) tab[i] = (unsigned short)((i * 2654435761u) >> 20);
    __syncthreads();
    const unsigned char *tb = (const unsigned char *)tab;
    unsigned state;
    asm volatile("v_mov_b32 %0, 0x130000" : "=v"(state));
    const unsigned thr = (f << 20) - 1u;
    const int n2 = -2 * (int)f;
    long long t0 = __builtin_readcyclecounter();
    if ((int)threadIdx.x < lanes) /* fewer active lanes: does a dependent instruction issue sooner? */
#pragma unroll 8
    for (int i = 0; i < iters; i++) {
        if (MODE == 0 || MODE == 1) {
            const unsigned x = state > thr ? state >> 16 : state;
            const unsigned q = __umulhi(x, mg);
            const unsigned at = (unsigned)__mul24((int)q, n2) + 2u * x;
            unsigned ent;
            if (MODE == 0) ent = *(const unsigned short *)(tb + (at & 0x3ffe));
            else ent = at & 0x1fff;
            state = (q << 12) + ent;
        } else if (MODE == 2) {
            state = *(const unsigned short *)(tb + ((state * 2u) & 0x3ffe));
        } else if (MODE == 3) {
            state = (state ^ (state >> 3)) + 0x9e37u;   // 3 dependent simple VALU ops
            state = (state ^ (state >> 5)) + 0x85ebu;
        } else if (MODE == 4) {
            state = __umulhi(state | 0x80000000u, mg) + 12345u; // mul_hi + add
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = state;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
    unsigned *out; long long *cyc, h;
    CK(hipMalloc(&out, 256)); CK(hipMalloc(&cyc, 8));
    const int iters = 200000;
    const char *names[] = {"full step (VALU chain + LDS lookup)", "step without LDS lookup", "dependent ds_read_u16 chain (+2 VALU)", "6 dependent simple VALU ops", "mul_hi_u32 + or + add"};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
#define RUNG(M, GRID) { hipLaunchKernelGGL(k<M>, dim3(GRID), dim3(64), 0, 0, out, cyc, iters, 977u, 0xFFFFFFFFu / 977u); CK(hipDeviceSynchronize()); \
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k<M>, dim3(GRID), dim3(64), 0, 0, out, cyc, iters, 977u, 0xFFFFFFFFu / 977u); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); \
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost)); \
    printf("%-42s grid %5d: %7.1f ticks/step, %7.2f ns/step, implied clock %.2f GHz\n", names[M], GRID, (double)h / iters, ms * 1e6 / iters, (double)h / (ms * 1e6)); }
#define RUN(M) RUNG(M, 1) RUNG(M, 1024)
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4)
    RUNG(0, 4096) RUNG(0, 8192)
    for (int lanes = 64; lanes >= 8; lanes >>= 1) {
#define RUNL(M) { hipLaunchKernelGGL(k<M>, dim3(1), dim3(64), 0, 0, out, cyc, iters, 977u, 0xFFFFFFFFu / 977u, lanes); CK(hipDeviceSynchronize()); \
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k<M>, dim3(1), dim3(64), 0, 0, out, cyc, iters, 977u, 0xFFFFFFFFu / 977u, lanes); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); \
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); printf("lanes %2d  %-42s %7.2f ns/step\n", lanes, names[M], ms * 1e6 / iters); }
        RUNL(0) RUNL(2) RUNL(3) RUNL(4)
    }
    int clk = 0; CK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0));
    int wclk = 0; CK(hipDeviceGetAttribute(&wclk, hipDeviceAttributeWallClockRate, 0));
    printf("device clock %d kHz, wall clock (readcyclecounter) %d kHz\n", clk, wclk);
    return 0;
}
