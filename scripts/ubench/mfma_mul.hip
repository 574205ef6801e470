// Does v_mfma_f32_4x4x1_16b_f32 with C = 0 return the IEEE-rounded product of its operands, bit for bit, for every lane?
// D[i][j] of block b = A[i] * B[j]: lane (4b + j) supplies b_j and receives D[0..3][j] in four registers; lane (4b + i)
// supplies a_i.  If so, a lane's sample times four constants is one matrix instruction instead of four v_mul_f32.
// build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off scripts/ubench/mfma_mul.hip -o scripts/ubench/mfma_mul
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef float float4v __attribute__((ext_vector_type(4)));

__global__ void k_probe(const float *x, const float *c4 /* 4 constants */, float *out /* [n][4] */, float *ref /* [n][4] */) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const float a = c4[lane & 3]; /* lane 4b + i supplies a_i = constant i */
    const float b = x[t];         /* lane 4b + j supplies b_j = its own sample */
    float4v acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc, 0, 0, 0);
    for (int i = 0; i < 4; i++) {
        out[t * 4 + i] = acc[i];
        float p;
        asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p) : "v"(c4[i]), "v"(b));
        ref[t * 4 + i] = p;
    }
}

int main() {
    const int n = 1 << 20;
    float *hx = (float *)malloc(n * 4), *ho = (float *)malloc(n * 16), *hr = (float *)malloc(n * 16);
    uint32_t s = 12345u;
    for (int i = 0; i < n; i++) {
        s = s * 1664525u + 1013904223u;
        uint32_t bits;
        switch (i & 7) {
        case 0: bits = s; break;                                             /* any bit pattern (incl. NaN, inf, denormals) */
        case 1: bits = (s & 0x007FFFFFu) | 0x3F000000u; break;               /* [0.5, 1) */
        case 2: bits = (s & 0x807FFFFFu) | 0x3C000000u; break;               /* +-[2^-7, 2^-6) */
        case 3: bits = (s & 0x807FFFFFu) | ((s >> 9 & 15u) + 100u) << 23; break; /* small normals */
        case 4: bits = s & 0x807FFFFFu; break;                               /* denormals and zeros */
        default: { float v = (float)(int)(s >> 5) * (1.0f / 134217728.0f); memcpy(&bits, &v, 4); } /* multiples of 2^-27 below 1 */
        }
        memcpy(&hx[i], &bits, 4);
    }
    const float sets[3][4] = {{(float)0.17338, (float)0.146984, (float)0.0982119, (float)0.0344874},
                              {(float)0.16332, (float)0.0676495, 0.125f, -0.125f},
                              {-(float)0.17338, 1969.0f, 5.0f, 0.0f}};
    float *dx, *dc, *dout, *dref;
    hipMalloc(&dx, n * 4);
    hipMalloc(&dc, 16);
    hipMalloc(&dout, n * 16);
    hipMalloc(&dref, n * 16);
    hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
    long bad = 0, nan_pairs = 0, zero_sign = 0, denorm_in = 0;
    for (int k = 0; k < 3; k++) {
        hipMemcpy(dc, sets[k], 16, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_probe, dim3(n / 256), dim3(256), 0, 0, dx, dc, dout, dref);
        hipMemcpy(ho, dout, n * 16, hipMemcpyDeviceToHost);
        hipMemcpy(hr, dref, n * 16, hipMemcpyDeviceToHost);
        for (long i = 0; i < (long)n * 4; i++) {
            uint32_t a, b;
            memcpy(&a, &ho[i], 4);
            memcpy(&b, &hr[i], 4);
            if (a == b)
                continue;
            if (ho[i] != ho[i] && hr[i] != hr[i]) { nan_pairs++; continue; }
            if ((a | b) == 0x80000000u) { zero_sign++; continue; }  /* +0 vs -0 */
            uint32_t xb;
            memcpy(&xb, &hx[i / 4], 4);
            if ((xb & 0x7F800000u) == 0) { denorm_in++; continue; }
            if (bad < 10)
                printf("MISMATCH set %d x=%08x c=%g mfma=%08x mul=%08x\n", k, xb, sets[k][i & 3], a, b);
            bad++;
        }
    }
    printf("%ld products compared: %ld mismatches (besides %ld NaN payload pairs, %ld zero-sign differences, %ld with denormal input)\n",
           (long)n * 12, bad, nan_pairs, zero_sign, denorm_in);
    return bad ? 1 : 0;
}
