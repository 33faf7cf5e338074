/*
 * TEST INFRASTRUCTURE ONLY -- drives the REAL libjpeg (third-party; the copy
 * found at build time, IJG libjpeg 9d in this image) with the same call
 * sequence the reference uses, to produce golden vectors and a CPU baseline.
 *
 *   MJPEG path : server/mjpeg_encoder.c:771-823 (setup), :830-855 (scanlines),
 *                :857-884 (finish); pixel converters :236-259
 *   still path : server/jpeg_encoder.c:232-246 with converters :107-158
 *
 * One deliberate knob: `fancy`.  The reference of that era linked the distro
 * libjpeg(-turbo), whose h2v2 chroma downsampling is the plain 2x2 box of
 * libjpeg v6b.  IJG 9 defaults to a different ("fancy") downsampler; passing
 * fancy=0 sets cinfo.do_fancy_downsampling = FALSE to obtain v6b behaviour.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <jpeglib.h>

enum { REF_FMT_16BIT = 6, REF_FMT_24BIT = 7, REF_FMT_32BIT = 8, REF_FMT_RGBA = 9 };

typedef struct RefMJpeg {
    struct jpeg_compress_struct cinfo;
    struct jpeg_error_mgr jerr;
    int first_frame;
    uint8_t *row;
    size_t row_size;
} RefMJpeg;

static void px32(const uint8_t *src, uint8_t *dst)
{
    uint32_t p;
    memcpy(&p, src, 4);
    dst[0] = (p >> 16) & 0xff;
    dst[1] = (p >> 8) & 0xff;
    dst[2] = p & 0xff;
}

static void px24(const uint8_t *src, uint8_t *dst)
{
    dst[0] = src[2];
    dst[1] = src[1];
    dst[2] = src[0];
}

static void px16(const uint8_t *src, uint8_t *dst)
{
    uint16_t p;
    memcpy(&p, src, 2);
    dst[0] = ((p >> 7) & 0xf8) | ((p >> 12) & 0x7);
    dst[1] = ((p >> 2) & 0xf8) | ((p >> 7) & 0x7);
    dst[2] = ((p << 3) & 0xf8) | ((p >> 2) & 0x7);
}

RefMJpeg *jref_mjpeg_new(void)
{
    RefMJpeg *m = (RefMJpeg *)calloc(1, sizeof(*m));
    m->first_frame = 1;
    m->cinfo.err = jpeg_std_error(&m->jerr);
    jpeg_create_compress(&m->cinfo);
    return m;
}

void jref_mjpeg_free(RefMJpeg *m)
{
    if (!m) {
        return;
    }
    jpeg_destroy_compress(&m->cinfo);
    free(m->row);
    free(m);
}

/* One frame through start_frame / encode_scanline* / end_frame.
 * dct: 0 ISLOW, 1 IFAST (the reference's MJPEG path uses IFAST).
 * Returns the JFIF size, or -1 if out is too small / format unsupported. */
long jref_mjpeg_frame(RefMJpeg *m, const uint8_t *pixels, int width, int height, long stride,
                      int fmt, int quality, int dct, int fancy, uint8_t *out, size_t cap)
{
    void (*conv)(const uint8_t *, uint8_t *);
    int bpp, y;
    unsigned char *buf = NULL;
    unsigned long buf_len = 0;

    switch (fmt) {
    case REF_FMT_32BIT:
    case REF_FMT_RGBA:
        bpp = 4;
        conv = px32;
        break;
    case REF_FMT_24BIT:
        bpp = 3;
        conv = px24;
        break;
    case REF_FMT_16BIT:
        bpp = 2;
        conv = px16;
        break;
    default:
        return -1;
    }
    if (m->row_size < (size_t)width * 3) {
        m->row = (uint8_t *)realloc(m->row, (size_t)width * 3);
        m->row_size = (size_t)width * 3;
    }

    m->cinfo.in_color_space = JCS_RGB;
    m->cinfo.input_components = 3;
    jpeg_mem_dest(&m->cinfo, &buf, &buf_len);
    m->cinfo.image_width = (JDIMENSION)width;
    m->cinfo.image_height = (JDIMENSION)height;
    jpeg_set_defaults(&m->cinfo);
    m->cinfo.dct_method = dct ? JDCT_IFAST : JDCT_ISLOW;
#if JPEG_LIB_VERSION >= 70
    m->cinfo.do_fancy_downsampling = fancy ? TRUE : FALSE;
#else
    (void)fancy;
#endif
    jpeg_set_quality(&m->cinfo, quality, TRUE);
    jpeg_start_compress(&m->cinfo, m->first_frame);

    for (y = 0; y < height; y++) {
        const uint8_t *src = pixels + (long)y * stride;
        uint8_t *dst = m->row;
        JSAMPROW rp = m->row;
        int x;
        for (x = 0; x < width; x++) {
            conv(src, dst);
            src += bpp;
            dst += 3;
        }
        jpeg_write_scanlines(&m->cinfo, &rp, 1);
    }
    jpeg_finish_compress(&m->cinfo);
    m->first_frame = 0;

    if (buf_len > cap) {
        free(buf);
        return -1;
    }
    memcpy(out, buf, buf_len);
    free(buf);
    return (long)buf_len;
}

/* Stateless convenience: new encoder, one frame, destroy. */
long jref_encode(const uint8_t *pixels, int width, int height, long stride, int fmt, int quality,
                 int dct, int fancy, uint8_t *out, size_t cap)
{
    RefMJpeg *m = jref_mjpeg_new();
    long n = jref_mjpeg_frame(m, pixels, width, height, stride, fmt, quality, dct, fancy, out, cap);
    jref_mjpeg_free(m);
    return n;
}

int jref_lib_version(void)
{
    return JPEG_LIB_VERSION;
}
