/*
 * libhydrium/libhydrium.h — public C API of the MI355X build of libhydrium.
 *
 * Drop-in contract: this header declares the same nine functions, types, enumerators and version
 * macros as the reference's src/include/libhydrium/libhydrium.h (v0.6.0, lines 17-51 version,
 * 67-107 enums, 109-155 metadata, 165-314 functions), with identical C ABI, so a program built
 * against the reference links and runs against this library unchanged and receives the same
 * bytes.  Only the prose is ours.  GPU-specific additions live in <hydrium_amd.h>.
 */
#ifndef HYDRIUM_H_
#define HYDRIUM_H_

#include <stddef.h>
#include <stdint.h>

/* ---- version: 0x1MMMmmmppp ---- */
#define HYDRIUM_VERSION_MAJOR 0
#define HYDRIUM_VERSION_MINOR 6
#define HYDRIUM_VERSION_POINT 0

#define HYDRIUM_COMPUTE_VERSION(ma, mi, po) \
    (UINT64_C(0x1000000000) | ((uint64_t)(ma) << 24) | ((uint64_t)(mi) << 12) | ((uint64_t)(po)))
#define HYDRIUM_VERSION_INT \
    HYDRIUM_COMPUTE_VERSION(HYDRIUM_VERSION_MAJOR, HYDRIUM_VERSION_MINOR, HYDRIUM_VERSION_POINT)

#define HYD_STRINGIFY0(n) #n
#define HYD_STRINGIFY(n) HYD_STRINGIFY0(n)
#define HYDRIUM_VERSION_STRING \
    HYD_STRINGIFY(HYDRIUM_VERSION_MAJOR) "." HYD_STRINGIFY(HYDRIUM_VERSION_MINOR) "." HYD_STRINGIFY(HYDRIUM_VERSION_POINT)

#ifdef _WIN32
#ifdef HYDRIUM_INTERNAL_BUILD
#define HYDRIUM_EXPORT __declspec(dllexport)
#else
#define HYDRIUM_EXPORT __declspec(dllimport)
#endif
#elif defined(__GNUC__) || defined(__clang__)
#define HYDRIUM_EXPORT __attribute__((visibility("default")))
#else
#define HYDRIUM_EXPORT
#endif

/* Return codes.  Anything below HYD_ERROR_START is a failure; HYD_NEED_MORE_OUTPUT is not. */
typedef enum HYDStatusCode {
    HYD_OK = 0,
    HYD_DEFAULT = -1,            /* internal placeholder, never returned */
    HYD_NEED_MORE_OUTPUT = -2,   /* output buffer full: release it, provide another, call hyd_flush */
    HYD_NEED_MORE_INPUT = -3,
    HYD_ERROR_START = -10,       /* threshold only, never returned */
    HYD_NOMEM = -13,             /* host or device allocation failed */
    HYD_API_ERROR = -14,         /* the caller broke the API contract; see hyd_error_message_get */
    HYD_INTERNAL_ERROR = -15,    /* library or device failure (includes "no usable GPU") */
} HYDStatusCode;

typedef enum HYDSampleFormat {
    HYD_UINT8,    /* 0..255 full range */
    HYD_UINT16,   /* 0..65535 full range */
    HYD_FLOAT32,  /* nominal 0.0..1.0, values outside are out-of-gamut colours; must be finite */
} HYDSampleFormat;

typedef struct HYDImageMetadata {
    size_t width;   /* pixels, 1 .. 2^30 */
    size_t height;  /* pixels, 1 .. 2^30, width * height <= 2^40 */
    /* non-zero: samples are linear light; zero: sRGB transfer.  BT.709 primaries, D65 either way */
    int linear_light;
    /*
     * Tile size selectors: 0..3 give tiles of 256, 512, 1024, 2048 pixels in that direction, each
     * tile coded as a Frame of its own.  -1 in either field selects one-frame mode: the image is a
     * single Frame sent as 2048x2048 tiles (this is what the reference CLI uses by default, and the
     * mode in which this build keeps the GPU busiest).
     */
    int tile_size_shift_x;
    int tile_size_shift_y;
} HYDImageMetadata;

typedef struct HYDEncoder HYDEncoder;

/* New encoder, or NULL when out of memory.  No GPU work happens until the first tile. */
HYDRIUM_EXPORT HYDEncoder *hyd_encoder_new(void);

/* Destroys the encoder; NULL is accepted.  Host memory is released at once.  The encoder's GPU context
 * (device buffers, pinned staging, streams) is kept parked for the next encoder of the same image shape in
 * this process — bounded in number and in megabytes, see hydamd_trim_cache() in hydrium_amd.h, which
 * releases what is parked. */
HYDRIUM_EXPORT HYDStatusCode hyd_encoder_destroy(HYDEncoder *encoder);

/* Must precede the first tile. */
HYDRIUM_EXPORT HYDStatusCode hyd_set_metadata(HYDEncoder *encoder, const HYDImageMetadata *metadata);

/*
 * Lends the encoder a buffer (at least 64 bytes) for codestream bytes.  Only one buffer may be on
 * loan at a time; it stays on loan until hyd_release_output_buffer.
 */
HYDRIUM_EXPORT HYDStatusCode hyd_provide_output_buffer(HYDEncoder *encoder, uint8_t *buffer, size_t buffer_len);

/*
 * Encodes one tile.
 *
 *   buffer[0..2]   first red, green and blue sample of the tile; the three may point into one
 *                  interleaved array (pixel_stride 3) or into separate planes (pixel_stride 1)
 *   tile_x, tile_y tile coordinates in units of tiles, raster order from the top left
 *   row_stride     distance between vertically adjacent samples, in SAMPLES (may be negative)
 *   pixel_stride   distance between horizontally adjacent samples of one channel, in SAMPLES
 *   is_last        1 / 0 to say whether this is the final tile, negative to let the library
 *                  assume the bottom-right tile is the final one
 *   sample_fmt     may differ from tile to tile
 *
 * The pixel memory is only borrowed for the duration of the call.  Edge tiles are clipped to the
 * image size given in the metadata, so row_stride must describe the caller's real row pitch.
 * After each tile call hyd_flush until it stops returning HYD_NEED_MORE_OUTPUT, swapping output
 * buffers in between.  In one-frame mode nothing but the file header is produced before the
 * final tile.  In tile mode (tile_size_shift >= 0) every tile is a frame of its own, complete when
 * the call returns — unless the caller opted into pipelined tile frames (hydamd_set_tile_pipeline in
 * hydrium_amd.h, or HYDAMD_TILE_PIPELINE=2..8 in the environment): then a tile's bytes may appear up
 * to depth - 1 calls after the call that sent it (always in send order; the call that sends the
 * final tile delivers everything still outstanding).
 * Returns HYD_OK or an error code.
 */
HYDRIUM_EXPORT HYDStatusCode hyd_send_tile(HYDEncoder *encoder, const void *const buffer[3],
                                           uint32_t tile_x, uint32_t tile_y, ptrdiff_t row_stride,
                                           ptrdiff_t pixel_stride, int is_last, HYDSampleFormat sample_fmt);

/* Takes the output buffer back; *written receives the number of valid bytes in it. */
HYDRIUM_EXPORT HYDStatusCode hyd_release_output_buffer(HYDEncoder *encoder, size_t *written);

/*
 * Moves pending codestream bytes into the buffer on loan.  HYD_NEED_MORE_OUTPUT means the buffer
 * is full and more bytes are pending; HYD_OK means everything produced so far has been delivered.
 */
HYDRIUM_EXPORT HYDStatusCode hyd_flush(HYDEncoder *encoder);

/* Static description of the most recent error, or NULL. */
HYDRIUM_EXPORT const char *hyd_error_message_get(HYDEncoder *encoder);

/*
 * Attaches an ICC profile as the suggested output colour space (one-frame mode only, before the
 * first tile).  The data is copied.  Passing NULL and 0 removes a previously set profile.
 */
HYDRIUM_EXPORT HYDStatusCode hyd_set_suggested_icc_profile(HYDEncoder *encoder,
    const uint8_t *icc_data, size_t icc_size);

#endif /* HYDRIUM_H_ */
