/* A plain C99 caller of the drop-in API, written the way the reference's CLI drives libhydrium
 * (reference src/hydrium.c:275-286,402-479): one-frame mode, a 1 MiB output buffer cycled through
 * flush / release / provide.  Built by tests/test_c_client.py against include/libhydrium/libhydrium.h
 * and hydrium_amd/lib/libhydrium.so.0; prints the codestream's size and a 64-bit FNV-1a hash. */
#include <libhydrium/libhydrium.h>

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

int main(int argc, char **argv) {
    const size_t w = argc > 1 ? (size_t)atoi(argv[1]) : 300, h = argc > 2 ? (size_t)atoi(argv[2]) : 200;
    uint8_t *rgb = malloc(w * h * 3);
    uint8_t *out = malloc(1 << 20);
    if (!rgb || !out)
        return 2;
    uint32_t s = 12345;
    for (size_t i = 0; i < w * h * 3; i++) { /* smooth-ish deterministic content */
        s = s * 1664525u + 1013904223u;
        rgb[i] = (uint8_t)(((i / 3) % w) / 2 + ((i / 3) / w) / 3 + (s >> 29));
    }
    HYDEncoder *enc = hyd_encoder_new();
    if (!enc)
        return 3;
    HYDImageMetadata md = {w, h, 0, -1, -1};
    HYDStatusCode st = hyd_set_metadata(enc, &md);
    if (st < HYD_ERROR_START)
        goto fail;
    st = hyd_provide_output_buffer(enc, out, 1 << 20);
    if (st < HYD_ERROR_START)
        goto fail;
    uint64_t hash = UINT64_C(0xcbf29ce484222325);
    size_t total = 0;
    const size_t ntx = (w + 2047) / 2048, nty = (h + 2047) / 2048;
    for (size_t ty = 0; ty < nty; ty++) {
        for (size_t tx = 0; tx < ntx; tx++) {
            const uint8_t *p = rgb + (ty * 2048 * w + tx * 2048) * 3;
            const void *const planes[3] = {p, p + 1, p + 2};
            st = hyd_send_tile(enc, planes, (uint32_t)tx, (uint32_t)ty, (ptrdiff_t)(3 * w), 3, -1, HYD_UINT8);
            if (st < HYD_ERROR_START)
                goto fail;
            do {
                st = hyd_flush(enc);
                if (st < HYD_ERROR_START)
                    goto fail;
                size_t n = 0;
                if (hyd_release_output_buffer(enc, &n) < HYD_ERROR_START)
                    goto fail;
                for (size_t i = 0; i < n; i++)
                    hash = (hash ^ out[i]) * UINT64_C(0x100000001b3);
                total += n;
                if (hyd_provide_output_buffer(enc, out, 1 << 20) < HYD_ERROR_START)
                    goto fail;
            } while (st == HYD_NEED_MORE_OUTPUT);
        }
    }
    hyd_encoder_destroy(enc);
    printf("%zu %016llx\n", total, (unsigned long long)hash);
    free(rgb);
    free(out);
    return 0;
fail:
    fprintf(stderr, "libhydrium error %d: %s\n", (int)st, hyd_error_message_get(enc));
    hyd_encoder_destroy(enc);
    return 1;
}
