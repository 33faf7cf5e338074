/*
 * C host side of the image compression path: the call sequences of
 * server/red_worker.c around the encoder libraries, over libspicegpu's drop-in
 * symbols.  See red_compress_host.h for the map to the reference.
 *
 * As in the reference, every encoder's callback struct is the FIRST member of
 * its data struct, so a callback downcasts `usr` to reach the buffer chain
 * and the line cursor (server/red_worker.c:561-604).
 */
#include "red_compress_host.h"

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#define MIN_GLZ_SIZE_FOR_ZLIB 100 /* server/red_worker.c:133 */
#define MAX_FREED 256

/* EncoderData, server/red_worker.c:561-580 */
typedef struct EncoderData {
    HostDisplayClient *dcc;
    RedCompressBuf *bufs_head, *bufs_tail;
    struct {
        const HostBitmap *bitmap;
        int next;
        uint32_t stride;
        int reverse;
    } lines;
    struct {
        RedCompressBuf *next;
        int size_left;
    } compressed;
} EncoderData;

typedef struct GlzData {
    GlzEncoderUsrContext usr; /* first */
    EncoderData data;
} GlzData;

typedef struct JpegData {
    JpegEncoderUsrContext usr; /* first */
    EncoderData data;
} JpegData;

struct HostDisplayClient {
    int image_compression;
    int enable_jpeg, jpeg_quality;
    int enable_zlib_glz_wrap, zlib_level;
    int migrate_freeze;
    GlzData glz_data;
    JpegData jpeg_data;
    EncoderData zlib_data;
    GlzEncDictContext *glz_dict;
    GlzEncoderContext *glz;
    JpegEncoderContext *jpeg;
    z_stream zstrm;
    int zlib_ready, zlib_last_level;
    RedCompressBuf *free_bufs, *used_bufs;
    uintptr_t freed[MAX_FREED];
    int nfreed;
    char error[256];
};

/* ---------------------------------------------------------------- buffers */
/* red_display_alloc_compress_buf, server/red_worker.c:5276-5300 */
static RedCompressBuf *alloc_compress_buf(HostDisplayClient *dcc)
{
    RedCompressBuf *buf = dcc->free_bufs;
    if (buf) {
        dcc->free_bufs = buf->next;
    } else {
        buf = (RedCompressBuf *)malloc(sizeof(*buf));
        if (!buf) {
            return NULL;
        }
    }
    buf->next = dcc->used_bufs;
    dcc->used_bufs = buf;
    buf->send_next = NULL;
    return buf;
}

void host_reset_compress_bufs(HostDisplayClient *dcc)
{
    while (dcc->used_bufs) {
        RedCompressBuf *buf = dcc->used_bufs;
        dcc->used_bufs = buf->next;
        buf->next = dcc->free_bufs;
        dcc->free_bufs = buf;
    }
}

/* encoder_usr_more_space, server/red_worker.c:5656-5668 (bytes for glz/jpeg/zlib) */
static int more_space(EncoderData *d, uint8_t **io_ptr)
{
    RedCompressBuf *buf = alloc_compress_buf(d->dcc);
    if (!buf) {
        return 0;
    }
    d->bufs_tail->send_next = buf;
    d->bufs_tail = buf;
    buf->send_next = NULL;
    *io_ptr = (uint8_t *)buf->buf;
    return (int)sizeof(buf->buf);
}

/* encoder_usr_more_lines, server/red_worker.c:5700-5728 */
static int more_lines(EncoderData *d, uint8_t **lines)
{
    const HostBitmap *bm = d->lines.bitmap;
    const HostChunk *chunk;
    if (d->lines.reverse) {
        if (d->lines.next < 0) {
            return 0;
        }
    } else if (d->lines.next >= (int)bm->num_chunks) {
        return 0;
    }
    chunk = &bm->chunk[d->lines.next];
    if (chunk->len % d->lines.stride) {
        return 0;
    }
    if (d->lines.reverse) {
        d->lines.next--;
        *lines = chunk->data + chunk->len - d->lines.stride;
    } else {
        d->lines.next++;
        *lines = chunk->data;
    }
    return (int)(chunk->len / d->lines.stride);
}

/* ---------------------------------------------------------- glz callbacks */
static void glz_usr_error(GlzEncoderUsrContext *usr, const char *fmt, ...)
{
    GlzData *g = (GlzData *)usr;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g->data.dcc->error, sizeof(g->data.dcc->error), fmt, ap);
    va_end(ap);
}

static void glz_usr_note(GlzEncoderUsrContext *usr, const char *fmt, ...)
{
    (void)usr;
    (void)fmt;
}

static void *glz_usr_malloc(GlzEncoderUsrContext *usr, int size)
{
    (void)usr;
    return malloc((size_t)size);
}

static void glz_usr_free(GlzEncoderUsrContext *usr, void *ptr)
{
    (void)usr;
    free(ptr);
}

static int glz_usr_more_space(GlzEncoderUsrContext *usr, uint8_t **io_ptr)
{
    return more_space(&((GlzData *)usr)->data, io_ptr);
}

static int glz_usr_more_lines(GlzEncoderUsrContext *usr, uint8_t **lines)
{
    return more_lines(&((GlzData *)usr)->data, lines);
}

/* glz_usr_free_image (server/red_worker.c:5773-5794) unlinks the drawable
 * instance; here the context is recorded for the caller */
static void glz_usr_free_image(GlzEncoderUsrContext *usr, GlzUsrImageContext *image)
{
    HostDisplayClient *dcc = ((GlzData *)usr)->data.dcc;
    if (dcc->nfreed < MAX_FREED) {
        dcc->freed[dcc->nfreed++] = (uintptr_t)image;
    }
}

static int jpeg_usr_more_space(JpegEncoderUsrContext *usr, uint8_t **io_ptr)
{
    return more_space(&((JpegData *)usr)->data, io_ptr);
}

static int jpeg_usr_more_lines(JpegEncoderUsrContext *usr, uint8_t **lines)
{
    return more_lines(&((JpegData *)usr)->data, lines);
}

/* -------------------------------------------------------------- life cycle */
HostDisplayClient *host_display_client_new(int image_compression, uint32_t glz_window_pixels,
                                           int enable_jpeg, int jpeg_quality,
                                           int enable_zlib_glz_wrap, int zlib_level)
{
    HostDisplayClient *dcc = (HostDisplayClient *)calloc(1, sizeof(*dcc));
    if (!dcc) {
        return NULL;
    }
    dcc->image_compression = image_compression;
    dcc->enable_jpeg = enable_jpeg;
    dcc->jpeg_quality = jpeg_quality;
    dcc->enable_zlib_glz_wrap = enable_zlib_glz_wrap;
    dcc->zlib_level = zlib_level;

    /* red_display_init_glz_data, server/red_worker.c:5578-5591 */
    dcc->glz_data.usr.error = glz_usr_error;
    dcc->glz_data.usr.warn = glz_usr_note;
    dcc->glz_data.usr.info = glz_usr_note;
    dcc->glz_data.usr.malloc = glz_usr_malloc;
    dcc->glz_data.usr.free = glz_usr_free;
    dcc->glz_data.usr.more_space = glz_usr_more_space;
    dcc->glz_data.usr.more_lines = glz_usr_more_lines;
    dcc->glz_data.usr.free_image = glz_usr_free_image;
    dcc->glz_data.data.dcc = dcc;
    dcc->jpeg_data.usr.more_space = jpeg_usr_more_space;
    dcc->jpeg_data.usr.more_lines = jpeg_usr_more_lines;
    dcc->jpeg_data.data.dcc = dcc;
    dcc->zlib_data.dcc = dcc;

    /* server/red_worker.c:9711 (one encoder here), :9635, :5848 */
    dcc->glz_dict = glz_enc_dictionary_create(glz_window_pixels, 1, &dcc->glz_data.usr);
    if (dcc->glz_dict) {
        dcc->glz = glz_encoder_create(0, dcc->glz_dict, &dcc->glz_data.usr);
    }
    dcc->jpeg = jpeg_encoder_create(&dcc->jpeg_data.usr);
    if (!dcc->glz_dict || !dcc->glz || !dcc->jpeg) {
        host_display_client_free(dcc);
        return NULL;
    }
    return dcc;
}

void host_display_client_free(HostDisplayClient *dcc)
{
    if (!dcc) {
        return;
    }
    if (dcc->jpeg) {
        jpeg_encoder_destroy(dcc->jpeg);
    }
    if (dcc->glz) {
        glz_encoder_destroy(dcc->glz);
    }
    if (dcc->glz_dict) {
        glz_enc_dictionary_destroy(dcc->glz_dict, &dcc->glz_data.usr);
    }
    if (dcc->zlib_ready) {
        deflateEnd(&dcc->zstrm);
    }
    host_reset_compress_bufs(dcc);
    while (dcc->free_bufs) {
        RedCompressBuf *buf = dcc->free_bufs;
        dcc->free_bufs = buf->next;
        free(buf);
    }
    free(dcc);
}

const char *host_last_error(const HostDisplayClient *dcc)
{
    return dcc->error;
}

int host_pop_freed_images(HostDisplayClient *dcc, uintptr_t *out, int max)
{
    int n = dcc->nfreed < max ? dcc->nfreed : max;
    memcpy(out, dcc->freed, (size_t)n * sizeof(out[0]));
    dcc->nfreed = 0;
    return n;
}

/* ------------------------------------------------------------------- zlib */
/* zlib_encode (server/zlib_encoder.c:68-125) with its two callbacks
 * (zlib_usr_more_input / _more_space, server/red_worker.c:5690-5698,
 * 5754-5771): the GLZ stream is read buffer by buffer from its chain */
static int zlib_wrap(HostDisplayClient *dcc, RedCompressBuf *glz_head, int glz_size,
                     RedCompressBuf **out_head)
{
    EncoderData *z = &dcc->zlib_data;
    int flush, enc_size = 0, out_size = 0, ret;

    if (!dcc->zlib_ready) {
        memset(&dcc->zstrm, 0, sizeof(dcc->zstrm));
        if (deflateInit(&dcc->zstrm, dcc->zlib_level) != Z_OK) {
            return -1;
        }
        dcc->zlib_ready = 1;
        dcc->zlib_last_level = dcc->zlib_level;
    }
    z->bufs_tail = alloc_compress_buf(dcc);
    z->bufs_head = z->bufs_tail;
    if (!z->bufs_head) {
        return -1;
    }
    z->compressed.next = glz_head;
    z->compressed.size_left = glz_size;

    if (deflateReset(&dcc->zstrm) != Z_OK) {
        return -1;
    }
    dcc->zstrm.next_out = (Bytef *)z->bufs_head->buf;
    dcc->zstrm.avail_out = sizeof(z->bufs_head->buf);
    if (dcc->zlib_level != dcc->zlib_last_level) {
        if (deflateParams(&dcc->zstrm, dcc->zlib_level, Z_DEFAULT_STRATEGY) != Z_OK) {
            return -1;
        }
        dcc->zlib_last_level = dcc->zlib_level;
    }
    do {
        /* zlib_usr_more_input */
        int buf_size;
        if (!z->compressed.next || z->compressed.size_left <= 0) {
            return -1;
        }
        buf_size = z->compressed.size_left < (int)sizeof(z->compressed.next->buf)
                       ? z->compressed.size_left
                       : (int)sizeof(z->compressed.next->buf);
        dcc->zstrm.next_in = (Bytef *)z->compressed.next->buf;
        dcc->zstrm.avail_in = (uInt)buf_size;
        z->compressed.next = z->compressed.next->send_next;
        z->compressed.size_left -= buf_size;

        enc_size += buf_size;
        flush = enc_size == glz_size ? Z_FINISH : Z_NO_FLUSH;
        for (;;) {
            int before = (int)dcc->zstrm.avail_out;
            uint8_t *p = NULL;
            ret = deflate(&dcc->zstrm, flush);
            if (ret == Z_STREAM_ERROR) {
                return -1;
            }
            out_size += before - (int)dcc->zstrm.avail_out;
            if (dcc->zstrm.avail_out) {
                break;
            }
            dcc->zstrm.avail_out = (uInt)more_space(z, &p);
            dcc->zstrm.next_out = p;
            if (dcc->zstrm.avail_out == 0) {
                return -1;
            }
        }
    } while (flush != Z_FINISH);
    *out_head = z->bufs_head;
    return ret == Z_STREAM_END ? out_size : -1;
}

/* -------------------------------------------------------------------- glz */
static int glz_compress_image(HostDisplayClient *dcc, const HostBitmap *src, uintptr_t usr_context,
                              HostCompressed *out)
{
    static const LzImageType fmt_to_lz[10] = {
        LZ_IMAGE_TYPE_INVALID, LZ_IMAGE_TYPE_INVALID, LZ_IMAGE_TYPE_INVALID, LZ_IMAGE_TYPE_INVALID,
        LZ_IMAGE_TYPE_INVALID, LZ_IMAGE_TYPE_INVALID, LZ_IMAGE_TYPE_RGB16, LZ_IMAGE_TYPE_RGB24,
        LZ_IMAGE_TYPE_RGB32, LZ_IMAGE_TYPE_RGBA};
    GlzData *g = &dcc->glz_data;
    RedCompressBuf *zhead = NULL;
    int glz_size, zlib_size;

    g->data.bufs_tail = alloc_compress_buf(dcc);
    g->data.bufs_head = g->data.bufs_tail;
    if (!g->data.bufs_head) {
        return 0;
    }
    g->data.lines.bitmap = src;
    g->data.lines.stride = src->stride;
    g->data.lines.next = 0;
    g->data.lines.reverse = 0;
    dcc->error[0] = 0;

    glz_size = glz_encode(dcc->glz, fmt_to_lz[src->format], (int)src->x, (int)src->y, src->top_down,
                          NULL, 0, (int)src->stride, (uint8_t *)g->data.bufs_head->buf,
                          sizeof(g->data.bufs_head->buf), (GlzUsrImageContext *)usr_context,
                          &out->glz_image);
    if (glz_size <= 0) {
        return 0; /* the reference's usr->error is fatal here (spice_critical) */
    }
    if (dcc->enable_zlib_glz_wrap && glz_size >= MIN_GLZ_SIZE_FOR_ZLIB) {
        zlib_size = zlib_wrap(dcc, g->data.bufs_head, glz_size, &zhead);
        if (zlib_size > 0 && zlib_size < glz_size) {
            out->type = HOST_IMAGE_TYPE_ZLIB_GLZ_RGB;
            out->glz_data_size = (uint32_t)glz_size;
            out->comp_buf = zhead;
            out->comp_buf_size = (uint32_t)zlib_size;
            return 1;
        }
    }
    out->type = HOST_IMAGE_TYPE_GLZ_RGB;
    out->comp_buf = g->data.bufs_head;
    out->comp_buf_size = (uint32_t)glz_size;
    return 1;
}

/* ------------------------------------------------------------------- jpeg */
static int jpeg_compress_image(HostDisplayClient *dcc, const HostBitmap *src, HostCompressed *out)
{
    JpegData *j = &dcc->jpeg_data;
    JpegEncoderImageType in_type;
    int has_alpha = 0, stride, jpeg_size;

    switch (src->format) {
    case SPICE_GPU_BITMAP_FMT_16BIT:
        in_type = JPEG_IMAGE_TYPE_RGB16;
        break;
    case SPICE_GPU_BITMAP_FMT_24BIT:
        in_type = JPEG_IMAGE_TYPE_BGR24;
        break;
    case SPICE_GPU_BITMAP_FMT_32BIT:
        in_type = JPEG_IMAGE_TYPE_BGRX32;
        break;
    case SPICE_GPU_BITMAP_FMT_RGBA:
        in_type = JPEG_IMAGE_TYPE_BGRX32;
        has_alpha = 1;
        break;
    default:
        return 0;
    }
    if (has_alpha) {
        /* JPEG_ALPHA is the JPEG followed by lz_encode(LZ_IMAGE_TYPE_XXXA) of the
         * alpha plane in the same chain (server/red_worker.c:6281-6300); lz_encode
         * is spice-common's, so the whole image is left to the caller rather than
         * handed out with half its data */
        out->type = HOST_IMAGE_TYPE_JPEG_ALPHA;
        return HOST_COMPRESS_DELEGATE;
    }
    j->data.bufs_tail = alloc_compress_buf(dcc);
    j->data.bufs_head = j->data.bufs_tail;
    if (!j->data.bufs_head) {
        return 0;
    }
    j->data.lines.bitmap = src;
    j->data.lines.stride = src->stride;
    if (src->top_down) {
        j->data.lines.next = 0;
        j->data.lines.reverse = 0;
        stride = (int)src->stride;
    } else {
        j->data.lines.next = (int)src->num_chunks - 1;
        j->data.lines.reverse = 1;
        stride = -(int)src->stride;
    }
    jpeg_size = jpeg_encode(dcc->jpeg, dcc->jpeg_quality, in_type, (int)src->x, (int)src->y, NULL, 0,
                            stride, (uint8_t *)j->data.bufs_head->buf, sizeof(j->data.bufs_head->buf));
    if (jpeg_size <= 0 || (uint32_t)jpeg_size > src->y * src->stride) {
        return 0; /* failed, or bigger than the original */
    }
    out->is_lossy = 1;
    out->type = HOST_IMAGE_TYPE_JPEG;
    out->comp_buf = j->data.bufs_head;
    out->comp_buf_size = (uint32_t)jpeg_size;
    return 1;
}

/* ---------------------------------------------------- red_compress_image */
int host_compress_image(HostDisplayClient *dcc, const HostBitmap *src, int can_lossy,
                        int drawable_graduality, uintptr_t glz_usr_context, HostCompressed *out)
{
    SpiceGpuCodecQuery q;
    SpiceGpuCodec codec;

    memset(out, 0, sizeof(*out));
    memset(&q, 0, sizeof(q));
    q.image_compression = dcc->image_compression;
    q.format = src->format;
    q.x = src->x;
    q.y = src->y;
    q.stride = src->stride;
    q.chunks_unstable = src->unstable;
    q.can_lossy = can_lossy;
    q.enable_jpeg = dcc->enable_jpeg;
    q.graduality = drawable_graduality;
    q.glz_dict_size = glz_enc_dictionary_get_size(dcc->glz_dict);
    q.glz_frozen = dcc->migrate_freeze;

    codec = spice_gpu_choose_codec(&q);
    if (codec == SPICE_GPU_CODEC_NEEDS_GRADUALITY) {
        /* _get_bitmap_graduality_level over the bitmap's chunks */
        const uint8_t *ptrs[SPICE_GPU_GRADUAL_MAX_CHUNKS];
        uint32_t rows[SPICE_GPU_GRADUAL_MAX_CHUNKS];
        uint32_t c;
        if (src->num_chunks > SPICE_GPU_GRADUAL_MAX_CHUNKS) {
            snprintf(dcc->error, sizeof(dcc->error), "bitmap of %u chunks", src->num_chunks);
            return 0;
        }
        for (c = 0; c < src->num_chunks; c++) {
            ptrs[c] = src->chunk[c].data;
            rows[c] = src->chunk[c].len / src->stride;
        }
        q.graduality = spice_gpu_bitmap_graduality_level(src->format, (int)src->x, (int)src->num_chunks,
                                                         ptrs, rows);
        if (q.graduality == BITMAP_GRADUAL_INVALID) {
            return 0;
        }
        codec = spice_gpu_choose_codec(&q);
    }
    switch (codec) {
    case SPICE_GPU_CODEC_QUIC:
        out->type = HOST_IMAGE_TYPE_QUIC; /* red_quic_compress_image: spice-common */
        return HOST_COMPRESS_DELEGATE;
    case SPICE_GPU_CODEC_LZ:
        out->type = HOST_IMAGE_TYPE_LZ_RGB; /* red_lz_compress_image: spice-common */
        return HOST_COMPRESS_DELEGATE;
    case SPICE_GPU_CODEC_JPEG:
        return jpeg_compress_image(dcc, src, out);
    case SPICE_GPU_CODEC_GLZ:
        return glz_compress_image(dcc, src, glz_usr_context, out);
    default:
        return 0;
    }
}

size_t host_gather_compressed(const HostCompressed *c, uint8_t *dst, size_t cap)
{
    const RedCompressBuf *buf = c->comp_buf;
    size_t left = c->comp_buf_size, done = 0;
    while (left && buf && done < cap) {
        size_t n = left < sizeof(buf->buf) ? left : sizeof(buf->buf);
        if (n > cap - done) {
            n = cap - done;
        }
        memcpy(dst + done, buf->buf, n);
        done += n;
        left -= n;
        buf = buf->send_next;
    }
    return done;
}

/* ---------------------------------------------------------------- streams */
#define HOST_MAX_FPS 30 /* agent->fps = MAX_FPS, server/red_worker.c:3036 */

struct HostStreamAgent {
    MJpegEncoder *mjpeg_encoder;
    int use_rate_control;
    uint32_t fps;
    uint64_t last_send_time;
    uint8_t *stream_outbuf; /* dcc->send_data.stream_outbuf, server/red_worker.c:10607 */
    size_t stream_outbuf_size;
};

HostStreamAgent *host_stream_agent_new(int rate_control, uint64_t starting_bit_rate,
                                       MJpegEncoderRateControlCbs *cbs, void *opaque)
{
    HostStreamAgent *a = (HostStreamAgent *)calloc(1, sizeof(*a));
    if (!a) {
        return NULL;
    }
    a->fps = HOST_MAX_FPS;
    a->use_rate_control = rate_control;
    /* red_display_create_stream, server/red_worker.c:3039-3051 */
    a->mjpeg_encoder = rate_control ? mjpeg_encoder_new(1, starting_bit_rate, cbs, opaque)
                                    : mjpeg_encoder_new(0, 0, NULL, NULL);
    if (!a->mjpeg_encoder) {
        free(a);
        return NULL;
    }
    return a;
}

void host_stream_agent_free(HostStreamAgent *a)
{
    if (!a) {
        return;
    }
    mjpeg_encoder_destroy(a->mjpeg_encoder);
    free(a->stream_outbuf);
    free(a);
}

MJpegEncoder *host_stream_agent_encoder(HostStreamAgent *a)
{
    return a->mjpeg_encoder;
}

/* red_get_image_line, server/red_worker.c:8315-8339 */
static uint8_t *image_line(const HostBitmap *bm, size_t *offset, uint32_t *chunk_nr, uint32_t stride)
{
    const HostChunk *chunk = &bm->chunk[*chunk_nr];
    uint8_t *ret;
    if (*offset == chunk->len) {
        if (*chunk_nr == bm->num_chunks - 1) {
            return NULL;
        }
        *offset = 0;
        (*chunk_nr)++;
        chunk = &bm->chunk[*chunk_nr];
    }
    if (chunk->len - *offset < stride) {
        return NULL;
    }
    ret = chunk->data + *offset;
    *offset += stride;
    return ret;
}

long host_stream_frame(HostStreamAgent *a, const HostBitmap *image, int stream_top_down, int left,
                       int top, int right, int bottom, uint32_t mm_time, uint64_t now_ns,
                       const uint8_t **data)
{
    const int width = right - left, height = bottom - top;
    size_t offset = 0, n;
    uint32_t chunk = 0;
    int i, skip;

    *data = NULL;
    if (!a->use_rate_control && now_ns - a->last_send_time < 1000000000ull / a->fps) {
        return 0;
    }
    switch (mjpeg_encoder_start_frame(a->mjpeg_encoder, image->format, width, height,
                                      &a->stream_outbuf, &a->stream_outbuf_size, mm_time)) {
    case MJPEG_ENCODER_FRAME_DROP:
        return 0;
    case MJPEG_ENCODER_FRAME_ENCODE_START:
        break;
    default:
        return -1;
    }
    /* encode_frame, server/red_worker.c:8341-8377 */
    skip = stream_top_down ? top : (int)image->y - bottom;
    for (i = 0; i < skip; i++) {
        image_line(image, &offset, &chunk, image->stride);
    }
    for (i = 0; i < height; i++) {
        uint8_t *line = image_line(image, &offset, &chunk, image->stride);
        if (!line) {
            return -1;
        }
        line += (size_t)left * mjpeg_encoder_get_bytes_per_pixel(a->mjpeg_encoder);
        if (mjpeg_encoder_encode_scanline(a->mjpeg_encoder, line, (size_t)width) == 0) {
            return -1;
        }
    }
    n = mjpeg_encoder_end_frame(a->mjpeg_encoder);
    if (!n) {
        return -1;
    }
    a->last_send_time = now_ns;
    *data = a->stream_outbuf;
    return (long)n;
}
