/* configs[4] as a C program would run it: T threads, each encoding its share of a batch of raw RGB8 frames through the
 * drop-in API (hyd_encoder_new .. hyd_send_tile .. hyd_flush per frame, one 16 MiB output buffer per thread) — the rate
 * of the library without a Python harness in the loop (bench.py's batch_4k leg makes a dozen ctypes calls per tile under
 * the GIL).  usage: batch_client <raw rgb8 file> <width> <height> <threads> <frames>     (scripts/batch_client.py)
 * prints: frames, seconds, frames/s, and the FNV-1a hash of frame 0's codestream. */
#define _POSIX_C_SOURCE 200809L
#include <libhydrium/libhydrium.h>

#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

static const uint8_t *g_rgb;
static size_t g_w, g_h;
static int g_threads, g_frames;
static uint64_t g_hash0;
static size_t g_size0;

static int encode(uint8_t *out, size_t cap, size_t *size) {
    HYDEncoder *enc = hyd_encoder_new();
    if (!enc)
        return -1;
    HYDImageMetadata md = {g_w, g_h, 0, -1, -1};
    HYDStatusCode st = hyd_set_metadata(enc, &md);
    if (st >= HYD_ERROR_START)
        st = hyd_provide_output_buffer(enc, out, cap);
    size_t total = 0;
    const size_t ntx = (g_w + 2047) / 2048, nty = (g_h + 2047) / 2048;
    for (size_t ty = 0; ty < nty && st >= HYD_ERROR_START; ty++)
        for (size_t tx = 0; tx < ntx && st >= HYD_ERROR_START; tx++) {
            const uint8_t *p = g_rgb + (ty * 2048 * g_w + tx * 2048) * 3;
            const void *const planes[3] = {p, p + 1, p + 2};
            st = hyd_send_tile(enc, planes, (uint32_t)tx, (uint32_t)ty, (ptrdiff_t)(3 * g_w), 3, -1, HYD_UINT8);
            if (st < HYD_ERROR_START)
                break;
            st = hyd_flush(enc);
            if (st < HYD_ERROR_START)
                break;
            size_t n = 0;
            if (hyd_release_output_buffer(enc, &n) < HYD_ERROR_START || st == HYD_NEED_MORE_OUTPUT) {
                st = HYD_INTERNAL_ERROR; /* a 16 MiB buffer takes a 4K frame whole */
                break;
            }
            total += n;
            st = hyd_provide_output_buffer(enc, out + total, cap - total);
        }
    if (st < HYD_ERROR_START)
        fprintf(stderr, "encode failed: %s\n", hyd_error_message_get(enc));
    hyd_encoder_destroy(enc);
    *size = total;
    return st < HYD_ERROR_START ? -1 : 0;
}

static void *worker(void *arg) {
    const int t = (int)(intptr_t)arg;
    uint8_t *out = malloc(16u << 20);
    for (int f = t; f < g_frames; f += g_threads) {
        size_t n = 0;
        if (!out || encode(out, 16u << 20, &n))
            exit(3);
        if (f == 0) {
            uint64_t h = UINT64_C(0xcbf29ce484222325);
            for (size_t i = 0; i < n; i++)
                h = (h ^ out[i]) * UINT64_C(0x100000001b3);
            g_hash0 = h;
            g_size0 = n;
        }
    }
    free(out);
    return NULL;
}

int main(int argc, char **argv) {
    if (argc < 6)
        return 2;
    g_w = (size_t)atol(argv[2]);
    g_h = (size_t)atol(argv[3]);
    g_threads = atoi(argv[4]);
    g_frames = atoi(argv[5]);
    uint8_t *rgb = malloc(g_w * g_h * 3);
    FILE *f = fopen(argv[1], "rb");
    if (!rgb || !f || fread(rgb, 1, g_w * g_h * 3, f) != g_w * g_h * 3)
        return 2;
    fclose(f);
    g_rgb = rgb;
    pthread_t th[64];
    if (g_threads < 1 || g_threads > 64)
        return 2;
    for (int round = 0; round < 4; round++) { /* the first round creates and parks the device contexts */
        struct timespec a, b;
        clock_gettime(CLOCK_MONOTONIC, &a);
        for (int t = 0; t < g_threads; t++)
            pthread_create(&th[t], NULL, worker, (void *)(intptr_t)t);
        for (int t = 0; t < g_threads; t++)
            pthread_join(th[t], NULL);
        clock_gettime(CLOCK_MONOTONIC, &b);
        const double s = (double)(b.tv_sec - a.tv_sec) + 1e-9 * (double)(b.tv_nsec - a.tv_nsec);
        printf("round %d: %d frames on %d threads in %.4f s = %.1f frames/s; frame 0: %zu bytes %016llx\n", round, g_frames,
               g_threads, s, g_frames / s, g_size0, (unsigned long long)g_hash0);
    }
    free(rgb);
    return 0;
}
