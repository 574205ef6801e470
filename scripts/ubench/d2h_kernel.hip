// How fast can a finished codestream (51 MB) reach pinned host memory, and by whom?
//   hipMemcpyAsync (whatever engine the runtime picks) against a copy kernel storing straight into mapped host memory,
//   with the GPU idle and with every CU busy (a VALU-bound kernel on another stream, like the transform kernel).
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/d2h_kernel scripts/ubench/d2h_kernel.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_copy(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = src[i];
}
__global__ void k_busy(float *out, int iters) {
    float a = threadIdx.x * 0.5f, b = 1.0001f;
    for (int i = 0; i < iters; i++) {
        a = a * b + 0.25f;
        b = b * 0.99999f + 1e-6f;
    }
    if (a == 12345.f)
        out[0] = a + b;
}

int main() {
    const size_t bytes = 51u << 20, n = bytes / 16;
    uint4 *src, *dst;
    float *sink;
    CK(hipMalloc(&src, bytes));
    CK(hipMemset(src, 1, bytes));
    CK(hipMalloc(&sink, 64));
    CK(hipHostMalloc((void **)&dst, bytes, hipHostMallocDefault));
    hipStream_t sc, sb;
    int lo, hi;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithPriority(&sc, hipStreamNonBlocking, hi));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int busy = 0; busy < 2; busy++) {
        for (int mode = 0; mode < 6; mode++) {
            const int wgs = mode == 0 ? 0 : 8 << (2 * (mode - 1)); /* 8 32 128 512 2048 */
            CK(hipDeviceSynchronize());
            if (busy) /* ~20 ms of VALU work on every CU, 8 waves per SIMD */
                for (int k = 0; k < 4; k++)
                    hipLaunchKernelGGL(k_busy, dim3(256 * 8), dim3(256), 0, sb, sink, 1 << 20);
            CK(hipEventRecord(e0, sc));
            for (int r = 0; r < 4; r++) {
                if (!wgs)
                    CK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, sc));
                else
                    hipLaunchKernelGGL(k_copy, dim3(wgs), dim3(256), 0, sc, src, dst, n);
            }
            CK(hipEventRecord(e1, sc));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%s  %-22s %7.2f ms per 51 MB = %5.1f GB/s\n", busy ? "CUs busy" : "GPU idle", wgs ? "" : "hipMemcpyAsync", ms / 4,
                   bytes * 4 / (ms * 1e-3) / 1e9);
            if (wgs)
                printf("          copy kernel, %4d workgroups\n", wgs);
            CK(hipDeviceSynchronize());
        }
    }
    return 0;
}
