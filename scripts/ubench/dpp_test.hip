#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out) {
    unsigned v = threadIdx.x * 3 + 1;
    unsigned a = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x15F, 0xF, 0xF, false); /* row_newbcast:15 */
    unsigned b = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false); /* row_mirror */
    out[threadIdx.x] = a;
    out[64 + threadIdx.x] = b;
}
int main() {
    unsigned *d, h[128];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int ok = 1;
    for (int i = 0; i < 64; i++) {
        unsigned wa = ((i | 15)) * 3 + 1, wb = ((i & ~15) | (15 - (i & 15))) * 3 + 1;
        if (h[i] != wa || h[64 + i] != wb) { ok = 0; printf("lane %d: newbcast %u (want %u) mirror %u (want %u)\n", i, h[i], wa, h[64 + i], wb); }
    }
    printf(ok ? "DPP_OK\n" : "DPP_BAD\n");
    return 0;
}
