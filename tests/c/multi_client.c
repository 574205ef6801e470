/* A plain C99 caller of the additive multi-device call (include/hydrium_amd.h: hydamd_multi_create /
 * hydamd_encode_image_multi / hydamd_multi_result / hydamd_multi_read; csrc/host/multi.c): the picture tests/c/api_client.c
 * codes through hyd_send_tile, put into device memory with the HIP runtime's C API and coded as ONE frame on `shards`
 * contexts of device 0 (an aliased device list: every cross-context step, only the xGMI hop missing), the assembling shard
 * rotating over three frames.  Prints what api_client.c prints — the codestream's size and its 64-bit FNV-1a hash — so
 * tests/test_c_client.py can hold it to the REFERENCE's output for the same picture.  "Host stays in C." */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <hydrium_amd.h>

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

int main(int argc, char **argv) {
    const size_t w = argc > 1 ? (size_t)atoi(argv[1]) : 4296, h = argc > 2 ? (size_t)atoi(argv[2]) : 4168;
    const int shards = argc > 3 ? atoi(argv[3]) : 2;
    uint8_t *rgb = malloc(w * h * 3);
    if (!rgb || shards < 1 || shards > HYDAMD_MAX_PEERS)
        return 2;
    uint32_t s = 12345;
    for (size_t i = 0; i < w * h * 3; i++) { /* the content of api_client.c */
        s = s * 1664525u + 1013904223u;
        rgb[i] = (uint8_t)(((i / 3) % w) / 2 + ((i / 3) / w) / 3 + (s >> 29));
    }
    void *d_rgb = NULL;
    if (hipMalloc(&d_rgb, w * h * 3) != hipSuccess || hipMemcpy(d_rgb, rgb, w * h * 3, hipMemcpyHostToDevice) != hipSuccess) {
        fprintf(stderr, "no device memory\n");
        return 3;
    }
    int devices[HYDAMD_MAX_PEERS] = {0}, st = 0;
    HYDImageMetadata md = {w, h, 0, -1, -1};
    HydAmdMulti *m = hydamd_multi_create(shards, devices, &md, &st);
    if (!m) {
        fprintf(stderr, "hydamd_multi_create: %d\n", st);
        return 4;
    }
    const void *src[3 * HYDAMD_MAX_PEERS];
    for (int d = 0; d < shards; d++) /* every shard's buffer is the whole picture here: origin = its first pixel */
        for (int c = 0; c < 3; c++)
            src[3 * d + c] = (const uint8_t *)d_rgb + c;
    uint8_t *out = NULL;
    size_t len = 0, first_len = 0;
    uint64_t first_hash = 0;
    for (int frame = 0; frame < 3; frame++) {
        if ((st = hydamd_encode_image_multi(m, src, (ptrdiff_t)(3 * w), 3, HYD_UINT8, frame % shards)) != 0 ||
            (st = hydamd_multi_result(m, &len)) != 0)
            goto fail;
        uint8_t *bigger = realloc(out, len);
        if (!bigger)
            goto fail;
        out = bigger;
        if ((st = hydamd_multi_read(m, out, len)) != 0)
            goto fail;
        uint64_t hash = UINT64_C(0xcbf29ce484222325);
        for (size_t i = 0; i < len; i++)
            hash = (hash ^ out[i]) * UINT64_C(0x100000001b3);
        if (frame == 0) {
            first_len = len;
            first_hash = hash;
        } else if (len != first_len || hash != first_hash) {
            fprintf(stderr, "frame %d (assembled on shard %d) differs from frame 0\n", frame, frame % shards);
            return 5;
        }
    }
    printf("%zu %016llx\n", first_len, (unsigned long long)first_hash);
    hydamd_multi_destroy(m);
    (void)hipFree(d_rgb);
    free(out);
    free(rgb);
    return 0;
fail:
    fprintf(stderr, "hydamd_multi error %d: %s\n", st, hydamd_multi_error(m));
    hydamd_multi_destroy(m);
    return 1;
}
