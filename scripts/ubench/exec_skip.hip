// Does a VALU instruction with only lanes 0-15 active cost less issue time than a full wave64 one?
// Also: plain vs packed f32 mul/add rate.  Build: hipcc --offload-arch=gfx950 -O3 exec_skip.hip -o exec_skip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters, unsigned seed) {
    const int lane = threadIdx.x & 63;
    unsigned a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 + 9, a5 = a0 ^ 77, a6 = a0 + 123, a7 = a0 * 11;
    if (MODE == 0 || (MODE == 1 && lane < 16) || (MODE == 2 && lane < 32) || (MODE == 3 && (lane & 3) == 0)) {
        for (int i = 0; i < iters; i++) {
            // 8 independent integer chains: issue-bound, not latency-bound
            a0 = a0 * 0x10001u + 1u; a1 = a1 * 0x10003u + 3u; a2 = a2 * 0x10005u + 5u; a3 = a3 * 0x10007u + 7u;
            a4 = a4 * 0x10009u + 9u; a5 = a5 * 0x1000bu + 11u; a6 = a6 * 0x1000du + 13u; a7 = a7 * 0x1000fu + 15u;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

typedef float float2_ __attribute__((ext_vector_type(2)));
template <int PACKED>
__global__ __launch_bounds__(256) void kf(float *out, int iters, float seed) {
    float x = seed + threadIdx.x * 1e-3f;
    if (PACKED) {
        float2_ a = {x, x + 1}, b = {x + 2, x + 3}, c = {x + 4, x + 5}, d = {x + 6, x + 7};
        const float2_ m = {1.0001f, 0.9999f}, n = {0.5f, 0.25f};
        for (int i = 0; i < iters; i++) {
            a = a * m; a = a + n; b = b * m; b = b + n; c = c * m; c = c + n; d = d * m; d = d + n;
        }
        out[blockIdx.x * 256 + threadIdx.x] = a.x + a.y + b.x + b.y + c.x + c.y + d.x + d.y;
    } else {
        float a0 = x, a1 = x + 1, b0 = x + 2, b1 = x + 3, c0 = x + 4, c1 = x + 5, d0 = x + 6, d1 = x + 7;
        for (int i = 0; i < iters; i++) {
            a0 = a0 * 1.0001f; a0 = a0 + 0.5f; a1 = a1 * 0.9999f; a1 = a1 + 0.25f; b0 = b0 * 1.0001f; b0 = b0 + 0.5f; b1 = b1 * 0.9999f; b1 = b1 + 0.25f;
            c0 = c0 * 1.0001f; c0 = c0 + 0.5f; c1 = c1 * 0.9999f; c1 = c1 + 0.25f; d0 = d0 * 1.0001f; d0 = d0 + 0.5f; d1 = d1 * 0.9999f; d1 = d1 + 0.25f;
        }
        out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + b0 + b1 + c0 + c1 + d0 + d1;
    }
}

template <typename F> float timeit(F f) {
    hipEvent_t ev0, ev1; CK(hipEventCreate(&ev0)); CK(hipEventCreate(&ev1));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(ev0)); f(); CK(hipEventRecord(ev1)); CK(hipEventSynchronize(ev1));
    float ms; CK(hipEventElapsedTime(&ms, ev0, ev1)); return ms;
}

int main() {
    unsigned *out; float *outf;
    const int blocks = 256 * 8, iters = 20000;
    CK(hipMalloc(&out, blocks * 256 * 4)); CK(hipMalloc(&outf, blocks * 256 * 4));
    float t0 = timeit([&] { hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1u); });
    float t1 = timeit([&] { hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1u); });
    printf("int mad, all 64 lanes : %.3f ms\nint mad, lanes 0-15   : %.3f ms  (ratio %.2f)\n", t0, t1, t0 / t1);
    float t2 = timeit([&] { hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1u); });
    float t3 = timeit([&] { hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, out, iters, 1u); });
    printf("int mad, lanes 0-31   : %.3f ms  (ratio %.2f)\nint mad, every 4th lane: %.3f ms  (ratio %.2f)\n", t2, t0 / t2, t3, t0 / t3);
    float f0 = timeit([&] { hipLaunchKernelGGL(kf<0>, dim3(blocks), dim3(256), 0, 0, outf, iters, 1.0f); });
    float f1 = timeit([&] { hipLaunchKernelGGL(kf<1>, dim3(blocks), dim3(256), 0, 0, outf, iters, 1.0f); });
    const double ops = (double)blocks * 256 * iters * 16;
    printf("f32 mul+add scalar    : %.3f ms  (%.1f Tops/s)\nf32 mul+add packed    : %.3f ms  (%.1f Tops/s, ratio %.2f)\n", f0, ops / f0 / 1e9, f1, ops / f1 / 1e9, f0 / f1);
    return 0;
}
