/*
 * The reference's side of the boundary, in C: what red_worker.c does around
 * the encoder libraries when it compresses an image or a stream frame,
 * reduced to the state this path touches.  It calls libspicegpu through the
 * same symbols red_worker.c calls in spice-server (include/spice_gpu.h), with
 * the same callback structs, buffer chain and rules.
 *
 *   RedCompressBuf / encoder_usr_more_space / _more_lines   server/red_worker.c:504-509, 5656-5752
 *   red_glz_compress_image (+ zlib wrap)                    server/red_worker.c:6003-6102
 *   red_jpeg_compress_image                                 server/red_worker.c:6182-6279
 *   red_compress_image                                      server/red_worker.c:6411-6525
 *   marshaller_add_compressed                               server/red_worker.c:5221-5233
 *   encode_frame / red_marshall_stream_data                 server/red_worker.c:8315-8502
 */
#ifndef RED_COMPRESS_HOST_H
#define RED_COMPRESS_HOST_H

#include <stddef.h>
#include <stdint.h>

#include "spice_gpu.h"

#ifdef __cplusplus
extern "C" {
#endif

#define RED_COMPRESS_BUF_SIZE (1024 * 64) /* server/red_worker.c:503 */

typedef struct RedCompressBuf RedCompressBuf;
struct RedCompressBuf {
    uint32_t buf[RED_COMPRESS_BUF_SIZE / 4];
    RedCompressBuf *next;      /* owner list */
    RedCompressBuf *send_next; /* chain of one compressed image */
};

/* SpiceChunk / SpiceChunks / SpiceBitmap, the fields this path reads */
typedef struct HostChunk {
    uint8_t *data;
    uint32_t len;
} HostChunk;

typedef struct HostBitmap {
    int format;   /* SPICE_BITMAP_FMT_*: 6 16BIT, 7 24BIT, 8 32BIT, 9 RGBA */
    int top_down; /* SPICE_BITMAP_FLAGS_TOP_DOWN */
    int unstable; /* SPICE_CHUNKS_FLAGS_UNSTABLE */
    uint32_t x, y, stride;
    uint32_t num_chunks;
    HostChunk *chunk;
} HostBitmap;

/* SPICE_IMAGE_TYPE_* (spice-protocol's enums.h, which is not part of the
 * reference tree; the values are the protocol's) */
enum {
    HOST_IMAGE_TYPE_BITMAP = 0,
    HOST_IMAGE_TYPE_QUIC = 1,
    HOST_IMAGE_TYPE_LZ_RGB = 101,
    HOST_IMAGE_TYPE_GLZ_RGB = 102,
    HOST_IMAGE_TYPE_JPEG = 105,
    HOST_IMAGE_TYPE_ZLIB_GLZ_RGB = 107,
    HOST_IMAGE_TYPE_JPEG_ALPHA = 108
};

/* SpiceImage descriptor + compress_send_data_t (server/red_worker.c:5996-6001) */
typedef struct HostCompressed {
    int type;                 /* HOST_IMAGE_TYPE_* */
    RedCompressBuf *comp_buf; /* the image's bytes; NULL with HOST_COMPRESS_DELEGATE */
    uint32_t comp_buf_size;
    uint32_t glz_data_size;   /* ZLIB_GLZ_RGB: size of the GLZ stream inside */
    uint32_t jpeg_size;       /* JPEG_ALPHA: size of the JPEG part (never set: JPEG_ALPHA is delegated) */
    int is_lossy;
    GlzEncDictImageContext *glz_image;
} HostCompressed;

typedef struct HostDisplayClient HostDisplayClient; /* DisplayChannelClient + the RedWorker fields used here */

HostDisplayClient *host_display_client_new(int image_compression, uint32_t glz_window_pixels,
                                           int enable_jpeg, int jpeg_quality,
                                           int enable_zlib_glz_wrap, int zlib_level);
void host_display_client_free(HostDisplayClient *dcc);
const char *host_last_error(const HostDisplayClient *dcc);
/* GlzUsrImageContext values handed to free_image since the last call */
int host_pop_freed_images(HostDisplayClient *dcc, uintptr_t *out, int max);

/* red_compress_image.  Returns
 *   1  out holds a complete image of out->type (the reference's TRUE);
 *   0  the bitmap goes out uncompressed (the reference's FALSE);
 *   HOST_COMPRESS_DELEGATE  the reference would code the bitmap with an encoder
 *      of spice-common (QUIC, LZ, or JPEG plus the LZ-coded alpha plane), which
 *      this library does not replace: out->type names it, out carries no data,
 *      and the caller runs that encoder itself or sends the bitmap raw.  It is
 *      never success: a caller that only tests for non-zero must treat it as 0.
 * drawable_graduality: BITMAP_GRADUAL_INVALID to have it computed. */
#define HOST_COMPRESS_DELEGATE 2
int host_compress_image(HostDisplayClient *dcc, const HostBitmap *src, int can_lossy,
                        int drawable_graduality, uintptr_t glz_usr_context, HostCompressed *out);
/* marshaller_add_compressed: the chain's bytes in order */
size_t host_gather_compressed(const HostCompressed *c, uint8_t *dst, size_t cap);
/* red_display_reset_compress_buf (server/red_worker.c:9036): buffers of sent images return to the free list */
void host_reset_compress_bufs(HostDisplayClient *dcc);

/* one stream of one client: StreamAgent + red_marshall_stream_data */
typedef struct HostStreamAgent HostStreamAgent;
HostStreamAgent *host_stream_agent_new(int rate_control, uint64_t starting_bit_rate,
                                       MJpegEncoderRateControlCbs *cbs, void *opaque);
void host_stream_agent_free(HostStreamAgent *agent);
MJpegEncoder *host_stream_agent_encoder(HostStreamAgent *agent);
/* src rectangle (left, top, right, bottom) of the bitmap; returns the JFIF
 * size in *data (owned by the agent), 0 for a frame skipped or dropped, -1
 * where the reference returns FALSE */
long host_stream_frame(HostStreamAgent *agent, const HostBitmap *image, int stream_top_down,
                       int left, int top, int right, int bottom, uint32_t mm_time, uint64_t now_ns,
                       const uint8_t **data);

#ifdef __cplusplus
}
#endif
#endif
