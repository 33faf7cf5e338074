/*
 * TEST INFRASTRUCTURE ONLY -- see jpeg_oracle.h for scope and parity status.
 *
 * Restates, for 8-bit YCbCr 4:2:0 baseline with the standard tables:
 *   libjpeg jccolor.c   rgb_ycc_convert      (fixed-point colour conversion)
 *   libjpeg jcprepct.c  pre_process_data     (bottom-edge replication)
 *   libjpeg jcsample.c  fullsize/h2v2_downsample (+ right-edge replication)
 *   libjpeg jfdctint.c / jfdctfst.c          (ISLOW / IFAST forward DCT)
 *   libjpeg jcdctmgr.c  forward_DCT          (divisor tables, quantisation)
 *   libjpeg jccoefct.c  compress_data        (MCU order, dummy edge blocks)
 *   libjpeg jchuff.c    encode_one_block     (Huffman coding, 0xFF stuffing)
 *   libjpeg jcmarker.c / jcparam.c           (markers, quality scaling)
 * as configured by server/mjpeg_encoder.c:771-823 and server/jpeg_encoder.c:232-246.
 */
#include "jpeg_oracle.h"

#include <stdlib.h>
#include <string.h>

static const uint8_t STD_LUMA_Q[64] = {
    16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
    14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
    18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};

static const uint8_t STD_CHROMA_Q[64] = {
    17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
    24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};

static const uint8_t ZIGZAG[64] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
    12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
    58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

static const uint8_t DC_LUMA_BITS[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const uint8_t DC_CHROMA_BITS[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
static const uint8_t DC_VALS[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t AC_LUMA_BITS[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
static const uint8_t AC_LUMA_VALS[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61,
    0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52,
    0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25,
    0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45,
    0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64,
    0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99,
    0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6,
    0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3,
    0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8,
    0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
static const uint8_t AC_CHROMA_BITS[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
static const uint8_t AC_CHROMA_VALS[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61,
    0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33,
    0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18,
    0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44,
    0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63,
    0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a,
    0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97,
    0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4,
    0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca,
    0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7,
    0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

/* AA&N scale factors, libjpeg jcdctmgr.c (IFAST divisors) */
static const int16_t AAN_SCALES[64] = {
    16384, 22725, 21407, 19266, 16384, 12873, 8867,  4520,  22725, 31521, 29692, 26722, 22725,
    17855, 12299, 6270,  21407, 29692, 27969, 25172, 21407, 16819, 11585, 5906,  19266, 26722,
    25172, 22654, 19266, 15137, 10426, 5315,  16384, 22725, 21407, 19266, 16384, 12873, 8867,
    4520,  12873, 17855, 16819, 15137, 12873, 10114, 6967,  3552,  8867,  12299, 11585, 10426,
    8867,  6967,  4799,  2446,  4520,  6270,  5906,  5315,  4520,  3552,  2446,  1247};

typedef struct HuffTab {
    uint16_t code[256];
    uint8_t size[256];
} HuffTab;

typedef struct Sink {
    uint8_t *out;
    size_t pos, cap;
    int overflow;
    uint32_t acc;
    int nbits;
} Sink;

static void sink_byte(Sink *s, uint8_t b)
{
    if (s->pos >= s->cap) {
        s->overflow = 1;
        s->pos++;
        return;
    }
    s->out[s->pos++] = b;
}

static void sink_u16(Sink *s, unsigned v)
{
    sink_byte(s, (uint8_t)(v >> 8));
    sink_byte(s, (uint8_t)v);
}

/* jchuff.c emit_bits: MSB first, a zero byte stuffed after every 0xFF */
static void sink_bits(Sink *s, unsigned code, int size)
{
    if (size == 0) {
        return;
    }
    s->acc = (s->acc << size) | (code & ((1u << size) - 1));
    s->nbits += size;
    while (s->nbits >= 8) {
        uint8_t b = (uint8_t)(s->acc >> (s->nbits - 8));
        sink_byte(s, b);
        if (b == 0xff) {
            sink_byte(s, 0);
        }
        s->nbits -= 8;
    }
}

/* jchuff.c jpeg_make_c_derived_tbl */
static void build_huff(const uint8_t *bits, const uint8_t *vals, HuffTab *t)
{
    unsigned code = 0;
    int k = 0, l, i;
    memset(t, 0, sizeof(*t));
    for (l = 1; l <= 16; l++) {
        for (i = 0; i < bits[l - 1]; i++) {
            t->code[vals[k]] = (uint16_t)code;
            t->size[vals[k]] = (uint8_t)l;
            code++;
            k++;
        }
        code <<= 1;
    }
}

/* jcparam.c jpeg_quality_scaling + jpeg_add_quant_table(force_baseline) */
static void scaled_qtable(const uint8_t *base, int quality, uint16_t *q)
{
    int scale, i;
    if (quality <= 0) quality = 1;
    if (quality > 100) quality = 100;
    scale = quality < 50 ? 5000 / quality : 200 - quality * 2;
    for (i = 0; i < 64; i++) {
        long t = ((long)base[i] * scale + 50L) / 100L;
        if (t <= 0) t = 1;
        if (t > 255) t = 255;
        q[i] = (uint16_t)t;
    }
}

/* jcdctmgr.c start_pass_fdctmgr */
static void divisors(const uint16_t *q, int dct, int32_t *div)
{
    int i;
    for (i = 0; i < 64; i++) {
        if (dct == JPGO_DCT_ISLOW) {
            div[i] = (int32_t)q[i] << 3;
        } else {
            div[i] = (int32_t)(((int32_t)q[i] * (int32_t)AAN_SCALES[i] + (1 << 10)) >> 11);
        }
    }
}

/* ------------------------------------------------------------------- DCTs */
#define F_0_298631336 2446
#define F_0_390180644 3196
#define F_0_541196100 4433
#define F_0_765366865 6270
#define F_0_899976223 7373
#define F_1_175875602 9633
#define F_1_501321110 12299
#define F_1_847759065 15137
#define F_1_961570560 16069
#define F_2_053119869 16819
#define F_2_562915447 20995
#define F_3_072711026 25172
#define DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))

/* jfdctint.c (LL&M), CONST_BITS 13, PASS1_BITS 2; d holds samples - 128 */
static void fdct_islow(int32_t *d)
{
    int pass, i;
    for (pass = 0; pass < 2; pass++) {
        int step = pass ? 8 : 1, line = pass ? 1 : 8;
        int sh = pass ? 13 + 2 : 13 - 2;
        for (i = 0; i < 8; i++) {
            int32_t *p = d + i * line;
            int32_t t0 = p[0] + p[7 * step], t7 = p[0] - p[7 * step];
            int32_t t1 = p[1 * step] + p[6 * step], t6 = p[1 * step] - p[6 * step];
            int32_t t2 = p[2 * step] + p[5 * step], t5 = p[2 * step] - p[5 * step];
            int32_t t3 = p[3 * step] + p[4 * step], t4 = p[3 * step] - p[4 * step];
            int32_t t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
            int32_t z1, z2, z3, z4, z5;
            if (!pass) {
                p[0] = (t10 + t11) << 2;
                p[4 * step] = (t10 - t11) << 2;
            } else {
                p[0] = DESCALE(t10 + t11, 2);
                p[4 * step] = DESCALE(t10 - t11, 2);
            }
            z1 = (t12 + t13) * F_0_541196100;
            p[2 * step] = DESCALE(z1 + t13 * F_0_765366865, sh);
            p[6 * step] = DESCALE(z1 + t12 * (-F_1_847759065), sh);
            z1 = t4 + t7;
            z2 = t5 + t6;
            z3 = t4 + t6;
            z4 = t5 + t7;
            z5 = (z3 + z4) * F_1_175875602;
            t4 *= F_0_298631336;
            t5 *= F_2_053119869;
            t6 *= F_3_072711026;
            t7 *= F_1_501321110;
            z1 *= -F_0_899976223;
            z2 *= -F_2_562915447;
            z3 *= -F_1_961570560;
            z4 *= -F_0_390180644;
            z3 += z5;
            z4 += z5;
            p[7 * step] = DESCALE(t4 + z1 + z3, sh);
            p[5 * step] = DESCALE(t5 + z2 + z4, sh);
            p[3 * step] = DESCALE(t6 + z2 + z3, sh);
            p[1 * step] = DESCALE(t7 + z1 + z4, sh);
        }
    }
}

/* jfdctfst.c (AA&N), CONST_BITS 8, products truncated by an arithmetic shift */
#define FMUL(v, c) (((v) * (c)) >> 8)
static void fdct_ifast(int32_t *d)
{
    int pass, i;
    for (pass = 0; pass < 2; pass++) {
        int step = pass ? 8 : 1, line = pass ? 1 : 8;
        for (i = 0; i < 8; i++) {
            int32_t *p = d + i * line;
            int32_t t0 = p[0] + p[7 * step], t7 = p[0] - p[7 * step];
            int32_t t1 = p[1 * step] + p[6 * step], t6 = p[1 * step] - p[6 * step];
            int32_t t2 = p[2 * step] + p[5 * step], t5 = p[2 * step] - p[5 * step];
            int32_t t3 = p[3 * step] + p[4 * step], t4 = p[3 * step] - p[4 * step];
            int32_t t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
            int32_t z1, z2, z3, z4, z5, z11, z13;
            p[0] = t10 + t11;
            p[4 * step] = t10 - t11;
            z1 = FMUL(t12 + t13, 181);
            p[2 * step] = t13 + z1;
            p[6 * step] = t13 - z1;
            t10 = t4 + t5;
            t11 = t5 + t6;
            t12 = t6 + t7;
            z5 = FMUL(t10 - t12, 98);
            z2 = FMUL(t10, 139) + z5;
            z4 = FMUL(t12, 334) + z5;
            z3 = FMUL(t11, 181);
            z11 = t7 + z3;
            z13 = t7 - z3;
            p[5 * step] = z13 + z2;
            p[3 * step] = z13 - z2;
            p[1 * step] = z11 + z4;
            p[7 * step] = z11 - z4;
        }
    }
}

/* jcdctmgr.c forward_DCT: level shift, DCT, divide with rounding to nearest */
static void block_coefs(const uint8_t *plane, int pstride, int bx, int by, int dct,
                        const int32_t *div, int16_t *out)
{
    int32_t ws[64];
    int r, c, i;
    for (r = 0; r < 8; r++) {
        const uint8_t *row = plane + (size_t)(by * 8 + r) * (size_t)pstride + (size_t)bx * 8;
        for (c = 0; c < 8; c++) {
            ws[r * 8 + c] = (int32_t)row[c] - 128;
        }
    }
    if (dct == JPGO_DCT_ISLOW) {
        fdct_islow(ws);
    } else {
        fdct_ifast(ws);
    }
    for (i = 0; i < 64; i++) {
        int32_t q = div[i], t = ws[i];
        if (t < 0) {
            t = -t;
            t += q >> 1;
            t = (t >= q) ? t / q : 0;
            t = -t;
        } else {
            t += q >> 1;
            t = (t >= q) ? t / q : 0;
        }
        out[i] = (int16_t)t;
    }
}

/* ----------------------------------------------------- pixels to components */
typedef struct Planes {
    int mcu_cols, mcu_rows;
    int yb_w, yb_h; /* real luma blocks per row / column */
    int ystride, cstride;
    uint8_t *y, *cb, *cr;
} Planes;

static void px_to_rgb(const uint8_t *src, int fmt, uint8_t *rgb)
{
    if (fmt == JPGO_FMT_32BIT || fmt == JPGO_FMT_RGBA) {
        /* mjpeg_encoder.c:244-250 */
        rgb[0] = src[2];
        rgb[1] = src[1];
        rgb[2] = src[0];
    } else if (fmt == JPGO_FMT_24BIT) {
        /* mjpeg_encoder.c:236-242 */
        rgb[0] = src[2];
        rgb[1] = src[1];
        rgb[2] = src[0];
    } else {
        /* mjpeg_encoder.c:253-259 */
        uint16_t p = (uint16_t)(src[0] | (src[1] << 8));
        rgb[0] = (uint8_t)(((p >> 7) & 0xf8) | ((p >> 12) & 0x7));
        rgb[1] = (uint8_t)(((p >> 2) & 0xf8) | ((p >> 7) & 0x7));
        rgb[2] = (uint8_t)(((p << 3) & 0xf8) | ((p >> 2) & 0x7));
    }
}

static int fmt_bpp(int fmt)
{
    switch (fmt) {
    case JPGO_FMT_16BIT: return 2;
    case JPGO_FMT_24BIT: return 3;
    case JPGO_FMT_32BIT:
    case JPGO_FMT_RGBA: return 4;
    default: return 0;
    }
}

static int build_planes(const uint8_t *pixels, int width, int height, long stride, int fmt,
                        int color, Planes *pl)
{
    int bpp = fmt_bpp(fmt);
    int cb_w = (width + 15) / 16;         /* chroma blocks per row = MCU columns */
    int full_cw = cb_w * 16;              /* full-resolution chroma columns fed to h2v2 */
    int h2 = (height + 1) & ~1;           /* rows after the odd-row replication */
    int ch = h2 / 2;                      /* real downsampled rows */
    int x, y;
    uint8_t *fcb, *fcr;
    /* jccolor.c: 16-bit fixed point; two Cb terms differ between v6b and 9 */
    const int32_t r_y = 19595, g_y = 38470, b_y = 7471;
    const int32_t r_cb = color == JPGO_COLOR_IJG9 ? 11058 : 11059;
    const int32_t g_cb = color == JPGO_COLOR_IJG9 ? 21710 : 21709;
    const int32_t b_cb = 32768, r_cr = 32768, g_cr = 27439, b_cr = 5329;
    const int32_t half = 1 << 15, cbcr_off = 128 << 16;

    if (!bpp || width <= 0 || height <= 0) {
        return -1;
    }
    pl->mcu_cols = cb_w;
    pl->mcu_rows = (height + 15) / 16;
    pl->yb_w = (width + 7) / 8;
    pl->yb_h = (height + 7) / 8;
    pl->ystride = pl->mcu_cols * 16;
    pl->cstride = cb_w * 8;
    pl->y = (uint8_t *)malloc((size_t)pl->ystride * (size_t)pl->mcu_rows * 16);
    pl->cb = (uint8_t *)malloc((size_t)pl->cstride * (size_t)pl->mcu_rows * 8);
    pl->cr = (uint8_t *)malloc((size_t)pl->cstride * (size_t)pl->mcu_rows * 8);
    fcb = (uint8_t *)malloc((size_t)full_cw * (size_t)h2);
    fcr = (uint8_t *)malloc((size_t)full_cw * (size_t)h2);

    for (y = 0; y < h2; y++) {
        int sy = y < height ? y : height - 1; /* jcprepct.c: pad the row group */
        const uint8_t *src = pixels + (long)sy * stride;
        uint8_t *yrow = pl->y + (size_t)y * (size_t)pl->ystride;
        uint8_t *cbrow = fcb + (size_t)y * (size_t)full_cw;
        uint8_t *crrow = fcr + (size_t)y * (size_t)full_cw;
        for (x = 0; x < width; x++) {
            uint8_t rgb[3];
            int32_t r, g, b;
            px_to_rgb(src + (size_t)x * (size_t)bpp, fmt, rgb);
            r = rgb[0];
            g = rgb[1];
            b = rgb[2];
            yrow[x] = (uint8_t)((r_y * r + g_y * g + b_y * b + half) >> 16);
            cbrow[x] = (uint8_t)((-r_cb * r - g_cb * g + b_cb * b + cbcr_off + half - 1) >> 16);
            crrow[x] = (uint8_t)((r_cr * r - g_cr * g - b_cr * b + cbcr_off + half - 1) >> 16);
        }
        /* jcsample.c expand_right_edge */
        for (x = width; x < pl->ystride; x++) {
            yrow[x] = yrow[width - 1];
        }
        for (x = width; x < full_cw; x++) {
            cbrow[x] = cbrow[width - 1];
            crrow[x] = crrow[width - 1];
        }
    }
    /* luma rows below the image replicate the last row */
    for (y = h2; y < pl->mcu_rows * 16; y++) {
        memcpy(pl->y + (size_t)y * (size_t)pl->ystride,
               pl->y + (size_t)(h2 - 1) * (size_t)pl->ystride, (size_t)pl->ystride);
    }
    /* jcsample.c h2v2_downsample: 2x2 box, rounding bias alternates 1,2 */
    for (y = 0; y < ch; y++) {
        const uint8_t *a0 = fcb + (size_t)(2 * y) * (size_t)full_cw, *a1 = a0 + full_cw;
        const uint8_t *c0 = fcr + (size_t)(2 * y) * (size_t)full_cw, *c1 = c0 + full_cw;
        uint8_t *ob = pl->cb + (size_t)y * (size_t)pl->cstride;
        uint8_t *or_ = pl->cr + (size_t)y * (size_t)pl->cstride;
        for (x = 0; x < pl->cstride; x++) {
            int bias = (x & 1) ? 2 : 1;
            ob[x] = (uint8_t)((a0[2 * x] + a0[2 * x + 1] + a1[2 * x] + a1[2 * x + 1] + bias) >> 2);
            or_[x] = (uint8_t)((c0[2 * x] + c0[2 * x + 1] + c1[2 * x] + c1[2 * x + 1] + bias) >> 2);
        }
    }
    /* jcprepct.c: pad the downsampled data to a full iMCU row */
    for (y = ch; y < pl->mcu_rows * 8; y++) {
        memcpy(pl->cb + (size_t)y * (size_t)pl->cstride,
               pl->cb + (size_t)(ch - 1) * (size_t)pl->cstride, (size_t)pl->cstride);
        memcpy(pl->cr + (size_t)y * (size_t)pl->cstride,
               pl->cr + (size_t)(ch - 1) * (size_t)pl->cstride, (size_t)pl->cstride);
    }
    free(fcb);
    free(fcr);
    return 0;
}

static void free_planes(Planes *pl)
{
    free(pl->y);
    free(pl->cb);
    free(pl->cr);
}

/* jccoefct.c compress_data: six blocks of one MCU, dummy blocks at the edges */
static void mcu_coefs(const Planes *pl, int mx, int my, int dct, const int32_t *div_y,
                      const int32_t *div_c, int16_t *blk /* [6][64] */)
{
    int yi, xi;
    for (yi = 0; yi < 2; yi++) {
        int by = my * 2 + yi;
        for (xi = 0; xi < 2; xi++) {
            int bx = mx * 2 + xi;
            int16_t *o = blk + (yi * 2 + xi) * 64;
            if (by < pl->yb_h && bx < pl->yb_w) {
                block_coefs(pl->y, pl->ystride, bx, by, dct, div_y, o);
            } else {
                memset(o, 0, 64 * sizeof(int16_t));
                o[0] = (o - 64)[0]; /* DC of the block coded just before it */
            }
        }
    }
    block_coefs(pl->cb, pl->cstride, mx, my, dct, div_c, blk + 4 * 64);
    block_coefs(pl->cr, pl->cstride, mx, my, dct, div_c, blk + 5 * 64);
}

/* jchuff.c encode_one_block */
static int bit_length(int v)
{
    int n = 0;
    while (v) {
        n++;
        v >>= 1;
    }
    return n;
}

static void huff_block(Sink *s, const int16_t *blk, int last_dc, const HuffTab *dc,
                       const HuffTab *ac)
{
    int t = blk[0] - last_dc, t2 = t, n, k, r = 0;
    if (t < 0) {
        t = -t;
        t2--;
    }
    n = bit_length(t);
    sink_bits(s, dc->code[n], dc->size[n]);
    if (n) {
        sink_bits(s, (unsigned)t2, n);
    }
    for (k = 1; k < 64; k++) {
        t = blk[ZIGZAG[k]];
        if (t == 0) {
            r++;
            continue;
        }
        while (r > 15) {
            sink_bits(s, ac->code[0xf0], ac->size[0xf0]);
            r -= 16;
        }
        t2 = t;
        if (t < 0) {
            t = -t;
            t2--;
        }
        n = bit_length(t);
        sink_bits(s, ac->code[(r << 4) + n], ac->size[(r << 4) + n]);
        sink_bits(s, (unsigned)t2, n);
        r = 0;
    }
    if (r > 0) {
        sink_bits(s, ac->code[0], ac->size[0]);
    }
}

/* ----------------------------------------------------------------- markers */
static void put_dqt(Sink *s, int idx, const uint16_t *q)
{
    int i;
    sink_u16(s, 0xffdb);
    sink_u16(s, 67);
    sink_byte(s, (uint8_t)idx);
    for (i = 0; i < 64; i++) {
        sink_byte(s, (uint8_t)q[ZIGZAG[i]]);
    }
}

static void put_dht(Sink *s, int idx, const uint8_t *bits, const uint8_t *vals)
{
    int n = 0, i;
    for (i = 0; i < 16; i++) {
        n += bits[i];
    }
    sink_u16(s, 0xffc4);
    sink_u16(s, (unsigned)(2 + 1 + 16 + n));
    sink_byte(s, (uint8_t)idx);
    for (i = 0; i < 16; i++) {
        sink_byte(s, bits[i]);
    }
    for (i = 0; i < n; i++) {
        sink_byte(s, vals[i]);
    }
}

/* jcmarker.c: SOI, JFIF APP0, DQT x2, SOF0, DHT x4, SOS */
static void put_headers(Sink *s, int width, int height, const uint16_t *qy, const uint16_t *qc)
{
    static const uint8_t app0[] = {0xff, 0xe0, 0, 16, 'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0};
    size_t i;
    sink_u16(s, 0xffd8);
    for (i = 0; i < sizeof(app0); i++) {
        sink_byte(s, app0[i]);
    }
    put_dqt(s, 0, qy);
    put_dqt(s, 1, qc);
    sink_u16(s, 0xffc0);
    sink_u16(s, 17);
    sink_byte(s, 8);
    sink_u16(s, (unsigned)height);
    sink_u16(s, (unsigned)width);
    sink_byte(s, 3);
    sink_byte(s, 1);
    sink_byte(s, 0x22);
    sink_byte(s, 0);
    sink_byte(s, 2);
    sink_byte(s, 0x11);
    sink_byte(s, 1);
    sink_byte(s, 3);
    sink_byte(s, 0x11);
    sink_byte(s, 1);
    put_dht(s, 0x00, DC_LUMA_BITS, DC_VALS);
    put_dht(s, 0x10, AC_LUMA_BITS, AC_LUMA_VALS);
    put_dht(s, 0x01, DC_CHROMA_BITS, DC_VALS);
    put_dht(s, 0x11, AC_CHROMA_BITS, AC_CHROMA_VALS);
    sink_u16(s, 0xffda);
    sink_u16(s, 12);
    sink_byte(s, 3);
    sink_byte(s, 1);
    sink_byte(s, 0x00);
    sink_byte(s, 2);
    sink_byte(s, 0x11);
    sink_byte(s, 3);
    sink_byte(s, 0x11);
    sink_byte(s, 0);
    sink_byte(s, 63);
    sink_byte(s, 0);
}

long jpgo_header(int width, int height, int quality, uint8_t *out, size_t cap)
{
    Sink s;
    uint16_t qy[64], qc[64];
    memset(&s, 0, sizeof(s));
    s.out = out;
    s.cap = cap;
    scaled_qtable(STD_LUMA_Q, quality, qy);
    scaled_qtable(STD_CHROMA_Q, quality, qc);
    put_headers(&s, width, height, qy, qc);
    return s.overflow ? -1 : (long)s.pos;
}

long jpgo_coefficients(const uint8_t *pixels, int width, int height, long stride, int fmt,
                       int quality, int dct, int color, int16_t *coefs, size_t cap_mcus)
{
    Planes pl;
    uint16_t qy[64], qc[64];
    int32_t dy[64], dc[64];
    int mx, my;
    long n = 0;
    if (build_planes(pixels, width, height, stride, fmt, color, &pl)) {
        return -1;
    }
    scaled_qtable(STD_LUMA_Q, quality, qy);
    scaled_qtable(STD_CHROMA_Q, quality, qc);
    divisors(qy, dct, dy);
    divisors(qc, dct, dc);
    if ((size_t)pl.mcu_cols * (size_t)pl.mcu_rows > cap_mcus) {
        free_planes(&pl);
        return -1;
    }
    for (my = 0; my < pl.mcu_rows; my++) {
        for (mx = 0; mx < pl.mcu_cols; mx++) {
            mcu_coefs(&pl, mx, my, dct, dy, dc, coefs + n * 6 * 64);
            n++;
        }
    }
    free_planes(&pl);
    return n;
}

long jpgo_encode(const uint8_t *pixels, int width, int height, long stride, int fmt, int quality,
                 int dct, int color, uint8_t *out, size_t cap)
{
    Planes pl;
    Sink s;
    HuffTab dcl, acl, dcc, acc;
    uint16_t qy[64], qc[64];
    int32_t dy[64], dc[64];
    int last_dc[3] = {0, 0, 0};
    int mx, my, b;

    if (width > 65535 || height > 65535 || build_planes(pixels, width, height, stride, fmt, color, &pl)) {
        return -1;
    }
    memset(&s, 0, sizeof(s));
    s.out = out;
    s.cap = cap;
    scaled_qtable(STD_LUMA_Q, quality, qy);
    scaled_qtable(STD_CHROMA_Q, quality, qc);
    divisors(qy, dct, dy);
    divisors(qc, dct, dc);
    build_huff(DC_LUMA_BITS, DC_VALS, &dcl);
    build_huff(AC_LUMA_BITS, AC_LUMA_VALS, &acl);
    build_huff(DC_CHROMA_BITS, DC_VALS, &dcc);
    build_huff(AC_CHROMA_BITS, AC_CHROMA_VALS, &acc);

    put_headers(&s, width, height, qy, qc);
    for (my = 0; my < pl.mcu_rows; my++) {
        for (mx = 0; mx < pl.mcu_cols; mx++) {
            int16_t blk[6 * 64];
            mcu_coefs(&pl, mx, my, dct, dy, dc, blk);
            for (b = 0; b < 4; b++) {
                huff_block(&s, blk + b * 64, last_dc[0], &dcl, &acl);
                last_dc[0] = blk[b * 64];
            }
            huff_block(&s, blk + 4 * 64, last_dc[1], &dcc, &acc);
            last_dc[1] = blk[4 * 64];
            huff_block(&s, blk + 5 * 64, last_dc[2], &dcc, &acc);
            last_dc[2] = blk[5 * 64];
        }
    }
    /* jchuff.c flush_bits: pad the last byte with one-bits */
    if (s.nbits > 0) {
        sink_bits(&s, 0x7f, 7);
    }
    s.acc = 0;
    s.nbits = 0;
    sink_u16(&s, 0xffd9);
    free_planes(&pl);
    return s.overflow ? -1 : (long)s.pos;
}
