/*
 * Baseline JPEG (YCbCr 4:2:0, standard tables) for gfx950, bit-exact with the
 * libjpeg configuration the reference selects (server/mjpeg_encoder.c:771-823:
 * JDCT_IFAST; server/jpeg_encoder.c:232-246: JDCT_ISLOW).
 *
 * jpeg_strip_kernel: one 256-lane workgroup per group of four strips (a strip
 * is up to 10 MCUs of one MCU row, a group up to 40).
 *   phase 1  coalesced pixel reads (RGB16 / BGR24 / BGRX32 converters of
 *            server/mjpeg_encoder.c:236-259), libjpeg's fixed-point RGB->YCbCr,
 *            2x2 chroma box filter with the alternating 1,2 bias, edge
 *            replication exactly where libjpeg applies it; samples go to LDS
 *   phase 2  one lane per 8x8 block: the whole integer DCT (LL&M "islow" or
 *            AA&N "ifast") and the divide-with-rounding quantiser in
 *            registers; the coefficients go to LDS in zigzag order
 *   phase 3  Huffman, one wavefront per strip, one block at a time with a lane
 *            per zigzag position: the run before a coefficient comes from a
 *            ballot of "non-zero", code lengths are prefix-summed over the 64
 *            lanes, and every lane ORs its code into a small LDS ring that is
 *            flushed to the strip's slot as words complete
 * The DC predictor of the group's first MCU comes from recomputing the MCU
 * just before it (a one-MCU halo), so groups need no ordering.
 *
 * The three small kernels after it work on compressed bytes only: scan of
 * strip bit counts, bit-granular concatenation, 0xFF byte stuffing.
 */
#include <hip/hip_runtime.h>
#include <spice_target.h>

#include "jpeg_device.h"
#include "runtime.h"

#define JT JPEG_STRIP_THREADS
#define JG_MCUS (JPEG_STRIP_MCUS * JPEG_GROUP_STRIPS)
#define JSLOTS (JG_MCUS + 1)                  /* slot 0 is the halo MCU */
#define YT_STRIDE (JSLOTS * 16)               /* bytes per luma tile row */
#define CT_STRIDE (JSLOTS * 8)
#define JCOEF_STRIDE 33                       /* words per block in LDS: 64 int16 + 1 against bank conflicts */
#define JRING_WORDS 256                       /* per wavefront; a block adds at most 52 words */

#define JPEG_GLOBAL SPICE_GLOBAL
#define JPEG_CONST SPICE_CONST /* read through the scalar cache */
#define JPEG_FOUR_PER_CU SPICE_WAVES_PER_EU(4)

/* zigzag position -> natural index */
#define JZZ_LIST \
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, \
    13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, \
    52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63

/* where natural-order coefficient i goes in the zigzag sequence */
#define JIZZ_LIST \
    0, 1, 5, 6, 14, 15, 27, 28, 2, 4, 7, 13, 16, 26, 29, 42, 3, 8, 12, 17, 25, 30, 41, 43, 9, 11, 18, \
    24, 31, 40, 44, 53, 10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60, 21, 34, 37, \
    47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63

/* ---------------------------------------------------------------- pixels */
/* Source pixel (x, y) as stored, and its conversion to r | g << 8 | b << 16
 * (converters of mjpeg_encoder.c:236-259).  FMT: 6 x1r5g5b5, 7 b,g,r bytes,
 * 8 / 9 b,g,r,pad, JPEG_FMT_RGB24 r,g,b bytes.  The format is a template
 * parameter so that a lane's loads can all be in flight before the first
 * conversion waits for one. */
template <int FMT>
__device__ __forceinline__ uint32_t jpeg_fetch_raw(const JpegDevJob &job, int x, int y)
{
    const JPEG_GLOBAL uint8_t *row = (const JPEG_GLOBAL uint8_t *)(uintptr_t)job.pixels + (int64_t)y * job.stride;
    if (FMT == 8 || FMT == 9) {
        return ((const JPEG_GLOBAL uint32_t *)row)[x];
    } else if (FMT == 7 || FMT == JPEG_FMT_RGB24) {
        const JPEG_GLOBAL uint8_t *p = row + x * 3;
        return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
    } else {
        return ((const JPEG_GLOBAL uint16_t *)row)[x];
    }
}

template <int FMT>
__device__ __forceinline__ uint32_t jpeg_raw_to_rgb(uint32_t p)
{
    if (FMT == 8 || FMT == 9 || FMT == 7) {
        return ((p >> 16) & 0xff) | (p & 0xff00) | ((p & 0xff) << 16);
    } else if (FMT == JPEG_FMT_RGB24) {
        return p; /* already r,g,b: convert_RGB24_to_RGB24, server/jpeg_encoder.c:160-163 */
    } else {
        const uint32_t r = ((p >> 7) & 0xf8) | ((p >> 12) & 0x7);
        const uint32_t g = ((p >> 2) & 0xf8) | ((p >> 7) & 0x7);
        const uint32_t b = ((p << 3) & 0xf8) | ((p >> 2) & 0x7);
        return r | (g << 8) | (b << 16);
    }
}

template <int FMT>
__device__ __forceinline__ uint32_t jpeg_fetch_rgb(const JpegDevJob &job, int x, int y)
{
    return jpeg_raw_to_rgb<FMT>(jpeg_fetch_raw<FMT>(job, x, y));
}

/* libjpeg jccolor.c, 16-bit fixed point */
__device__ __forceinline__ int jpeg_y(uint32_t rgb)
{
    const int r = rgb & 0xff, g = (rgb >> 8) & 0xff, b = (rgb >> 16) & 0xff;
    return (19595 * r + 38470 * g + 7471 * b + 32768) >> 16;
}
__device__ __forceinline__ int jpeg_cb(uint32_t rgb)
{
    const int r = rgb & 0xff, g = (rgb >> 8) & 0xff, b = (rgb >> 16) & 0xff;
    return (-11059 * r - 21709 * g + 32768 * b + (128 << 16) + 32767) >> 16;
}
__device__ __forceinline__ int jpeg_cr(uint32_t rgb)
{
    const int r = rgb & 0xff, g = (rgb >> 8) & 0xff, b = (rgb >> 16) & 0xff;
    return (32768 * r - 27439 * g - 5329 * b + (128 << 16) + 32767) >> 16;
}

/* ------------------------------------------------------------------- DCTs */
#define JDESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))
/* Samples are 8 bits, so every DCT operand stays below 2^17 and every constant
 * below 2^15: the 24-bit multiplier gives the same low 32 bits at full rate. */
#define JMUL(a, b) __mul24((a), (b))

/* libjpeg jfdctint.c, one 8-point pass; FIRST: row pass (scale up by 4) */
template <bool FIRST>
__device__ __forceinline__ void jpeg_islow_1d(int &d0, int &d1, int &d2, int &d3, int &d4, int &d5,
                                              int &d6, int &d7)
{
    const int sh = FIRST ? 11 : 15;
    int t0 = d0 + d7, t7 = d0 - d7, t1 = d1 + d6, t6 = d1 - d6;
    int t2 = d2 + d5, t5 = d2 - d5, t3 = d3 + d4, t4 = d3 - d4;
    const int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
    if (FIRST) {
        d0 = (t10 + t11) << 2;
        d4 = (t10 - t11) << 2;
    } else {
        d0 = JDESCALE(t10 + t11, 2);
        d4 = JDESCALE(t10 - t11, 2);
    }
    int z1 = JMUL(t12 + t13, 4433);
    d2 = JDESCALE(z1 + JMUL(t13, 6270), sh);
    d6 = JDESCALE(z1 + JMUL(t12, -15137), sh);
    z1 = t4 + t7;
    int z2 = t5 + t6, z3 = t4 + t6, z4 = t5 + t7;
    const int z5 = JMUL(z3 + z4, 9633);
    t4 = JMUL(t4, 2446);
    t5 = JMUL(t5, 16819);
    t6 = JMUL(t6, 25172);
    t7 = JMUL(t7, 12299);
    z1 = JMUL(z1, -7373);
    z2 = JMUL(z2, -20995);
    z3 = JMUL(z3, -16069) + z5;
    z4 = JMUL(z4, -3196) + z5;
    d7 = JDESCALE(t4 + z1 + z3, sh);
    d5 = JDESCALE(t5 + z2 + z4, sh);
    d3 = JDESCALE(t6 + z2 + z3, sh);
    d1 = JDESCALE(t7 + z1 + z4, sh);
}

/* libjpeg jfdctfst.c, one 8-point pass (products truncated by >> 8) */
__device__ __forceinline__ void jpeg_ifast_1d(int &d0, int &d1, int &d2, int &d3, int &d4, int &d5,
                                              int &d6, int &d7)
{
    const int t0 = d0 + d7, t7 = d0 - d7, t1 = d1 + d6, t6 = d1 - d6;
    const int t2 = d2 + d5, t5 = d2 - d5, t3 = d3 + d4, t4 = d3 - d4;
    int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
    d0 = t10 + t11;
    d4 = t10 - t11;
    const int z1 = JMUL(t12 + t13, 181) >> 8;
    d2 = t13 + z1;
    d6 = t13 - z1;
    t10 = t4 + t5;
    t11 = t5 + t6;
    t12 = t6 + t7;
    const int z5 = JMUL(t10 - t12, 98) >> 8;
    const int z2 = (JMUL(t10, 139) >> 8) + z5;
    const int z4 = (JMUL(t12, 334) >> 8) + z5;
    const int z3 = JMUL(t11, 181) >> 8;
    const int z11 = t7 + z3, z13 = t7 - z3;
    d5 = z13 + z2;
    d3 = z13 - z2;
    d1 = z11 + z4;
    d7 = z11 - z4;
}

#define J8(a, o, s) a[o], a[o + s], a[o + 2 * s], a[o + 3 * s], a[o + 4 * s], a[o + 5 * s], a[o + 6 * s], a[o + 7 * s]
/* The same pass on two independent 8-point rows (or columns) at once, one in each half
 * of a pair of single-precision values (v_pk_add_f32 / v_pk_mul_f32).  Exact: the
 * operands are integers below 2^15, every product with a constant (at most 669) stays
 * below 2^24, c / 256 is a power-of-two scaling, and the shift by 8 of the integer code,
 * which rounds towards minus infinity, is the floor. */
typedef float jf2 __attribute__((vector_size(8)));

typedef float jf8 __attribute__((vector_size(32), aligned(4))); /* eight constants of a table row, wherever the row starts */

__device__ __forceinline__ uint32_t jpeg_fbits(float f)
{
    union { float f; uint32_t u; } x;
    x.f = f;
    return x.u;
}

__device__ __forceinline__ float jpeg_bitsf(uint32_t u)
{
    union { float f; uint32_t u; } x;
    x.u = u;
    return x.f;
}

__device__ __forceinline__ jf2 jpeg_f2(float a, float b)
{
    jf2 r;
    r[0] = a;
    r[1] = b;
    return r;
}

__device__ __forceinline__ jf2 jpeg_mulshift8(jf2 v, float c256)
{
    const jf2 m = v * jpeg_f2(c256, c256);
    return jpeg_f2(__builtin_floorf(m[0]), __builtin_floorf(m[1]));
}

__device__ __forceinline__ void jpeg_ifast_1d_f2(jf2 &d0, jf2 &d1, jf2 &d2, jf2 &d3, jf2 &d4, jf2 &d5,
                                                 jf2 &d6, jf2 &d7)
{
    const jf2 t0 = d0 + d7, t7 = d0 - d7, t1 = d1 + d6, t6 = d1 - d6;
    const jf2 t2 = d2 + d5, t5 = d2 - d5, t3 = d3 + d4, t4 = d3 - d4;
    jf2 t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
    d0 = t10 + t11;
    d4 = t10 - t11;
    const jf2 z1 = jpeg_mulshift8(t12 + t13, 181.0f / 256.0f);
    d2 = t13 + z1;
    d6 = t13 - z1;
    t10 = t4 + t5;
    t11 = t5 + t6;
    t12 = t6 + t7;
    const jf2 z5 = jpeg_mulshift8(t10 - t12, 98.0f / 256.0f);
    const jf2 z2 = jpeg_mulshift8(t10, 139.0f / 256.0f) + z5;
    const jf2 z4 = jpeg_mulshift8(t12, 334.0f / 256.0f) + z5;
    const jf2 z3 = jpeg_mulshift8(t11, 181.0f / 256.0f);
    const jf2 z11 = t7 + z3, z13 = t7 - z3;
    d5 = z13 + z2;
    d3 = z13 - z2;
    d1 = z11 + z4;
    d7 = z11 - z4;
}

/* jfdctfst.c on a block held as 32 pairs: on entry p[rp * 8 + k] = samples (2 rp, k) and
 * (2 rp + 1, k), unsigned (the level shift of 128 a sample reaches nothing but the DC
 * term -- every product is taken of a difference in which it cancels -- and is taken
 * off there, 64 x 128); on return p[r * 4 + kp] = coefficients (r, 2 kp) and (r, 2 kp + 1). */
__device__ __forceinline__ void jpeg_fdct_ifast_f2(jf2 (&p)[32])
{
#pragma unroll
    for (int rp = 0; rp < 4; rp++) {
        jpeg_ifast_1d_f2(J8(p, rp * 8, 1));
    }
    /* rows paired -> columns paired: a 2 x 2 transpose of every pair of pairs */
    jf2 q[32];
#pragma unroll
    for (int rp = 0; rp < 4; rp++) {
#pragma unroll
        for (int kp = 0; kp < 4; kp++) {
            const jf2 a = p[rp * 8 + 2 * kp], b = p[rp * 8 + 2 * kp + 1];
            q[(2 * rp) * 4 + kp] = jpeg_f2(a[0], b[0]);
            q[(2 * rp + 1) * 4 + kp] = jpeg_f2(a[1], b[1]);
        }
    }
#pragma unroll
    for (int kp = 0; kp < 4; kp++) {
        jpeg_ifast_1d_f2(J8(q, kp, 4));
    }
#pragma unroll
    for (int i = 0; i < 32; i++) {
        p[i] = q[i];
    }
    p[0][0] -= 8192.0f;
}

template <bool IFAST>
__device__ __forceinline__ void jpeg_fdct(int (&c)[64])
{
#pragma unroll
    for (int r = 0; r < 8; r++) {
        if (IFAST) {
            jpeg_ifast_1d(J8(c, r * 8, 1));
        } else {
            jpeg_islow_1d<true>(J8(c, r * 8, 1));
        }
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        if (IFAST) {
            jpeg_ifast_1d(J8(c, k, 8));
        } else {
            jpeg_islow_1d<false>(J8(c, k, 8));
        }
    }
}

/* ---------------------------------------------------------------- Huffman */
/* the n low bits of x (n < 32; none for n = 0): one bit-field extract */
__device__ __forceinline__ uint32_t jpeg_low_bits(uint32_t x, int n)
{
    return spice_low_bits(x, n);
}

/* packed table entry: size << 16 | code */
__device__ __forceinline__ int jpeg_nbits(int a)
{
    return 32 - __clz(a);
}

/* ORs the low `len` (1..32) bits of `code` into the bit stream at position
 * `pos` (MSB first inside 32-bit words of a ring of JRING_WORDS words). */
__device__ __forceinline__ void jpeg_ring_put(uint32_t *ring, uint32_t pos, uint32_t code, int len)
{
    const uint32_t w0 = pos >> 5;
    const uint64_t t = (uint64_t)code << (64 - (int)(pos & 31) - len);
    atomicOr(&ring[w0 & (JRING_WORDS - 1)], (uint32_t)(t >> 32));
    if ((uint32_t)t) {
        atomicOr(&ring[(w0 + 1) & (JRING_WORDS - 1)], (uint32_t)t);
    }
}

/* Exclusive scan of one int per lane over the workgroup; *total gets the sum. */
__device__ __forceinline__ int jpeg_block_scan(int v, int *wave_sums, int tid, int *total)
{
    const int lane = tid & 63, wave = tid >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d);
        if (lane >= d) {
            inc += o;
        }
    }
    if (lane == 63) {
        wave_sums[wave] = inc;
    }
    __syncthreads();
    int base = 0, sum = 0;
    for (int k = 0; k < JT / 64; k++) {
        const int s = wave_sums[k];
        if (k < wave) {
            base += s;
        }
        sum += s;
    }
    __syncthreads();
    *total = sum;
    return base + inc - v;
}

/* huff: [0..255] AC luma, [256..511] AC chroma, [512..527] DC luma, [528..543] DC chroma */
#define JHUFF_WORDS 544

/* Phase 1 of the strip kernel for one pixel format, a pixel per request: the
 * group's pixels (and its halo MCU's) become luma and 2x2-averaged chroma
 * samples in the LDS tiles. */
template <int FMT>
__device__ __forceinline__ void jpeg_load_tiles(const JpegDevJob &job, uint8_t *ytile, uint8_t *cbtile,
                                                uint8_t *crtile, int tid, int ns, bool has_halo, int hx,
                                                int hy, int mx0, int my)
{
    const int W = job.width, H = job.height;
    const int ch = (H + 1) >> 1; /* real downsampled rows */
    /* ---- phase 1: 2x2 pixel quads -> 4 luma samples + 1 Cb + 1 Cr.  All of a
     * lane's pixels are requested before the first is used. */
    const int quads_x = ns * 8;
    constexpr int JQ_ITERS = (JSLOTS * 64 + JT - 1) / JT;
    uint32_t px[JQ_ITERS][4];
#pragma unroll
    for (int it = 0; it < JQ_ITERS; it++) {
        const int item = tid + it * JT;
        px[it][0] = px[it][1] = px[it][2] = px[it][3] = 0;
        if (item < quads_x * 8) {
            const int qy = item / quads_x, qxs = item - qy * quads_x;
            const int slot = qxs >> 3, qx = qxs & 7;
            if (slot != 0 || has_halo) {
                const int mcx = slot == 0 ? hx : mx0 + slot - 1;
                const int mcy = slot == 0 ? hy : my;
                const int x = mcx * 16 + qx * 2;
                /* chroma: rows are replicated after downsampling (jcprepct.c), columns before (jcsample.c) */
                const int cyc = min(mcy * 8 + qy, ch - 1);
                const int r0 = min(2 * cyc, H - 1), r1 = min(2 * cyc + 1, H - 1);
                const int c0 = min(x, W - 1), c1 = min(x + 1, W - 1);
                px[it][0] = jpeg_fetch_raw<FMT>(job, c0, r0);
                px[it][1] = jpeg_fetch_raw<FMT>(job, c1, r0);
                px[it][2] = jpeg_fetch_raw<FMT>(job, c0, r1);
                px[it][3] = jpeg_fetch_raw<FMT>(job, c1, r1);
            }
        }
    }
#pragma unroll
    for (int it = 0; it < JQ_ITERS; it++) {
        const int item = tid + it * JT;
        if (item >= quads_x * 8) {
            continue;
        }
        const int qy = item / quads_x, qxs = item - qy * quads_x;
        const int slot = qxs >> 3, qx = qxs & 7;
        if (slot == 0 && !has_halo) {
            continue;
        }
        const int mcx = slot == 0 ? hx : mx0 + slot - 1;
        const int mcy = slot == 0 ? hy : my;
        const int x = mcx * 16 + qx * 2, y = mcy * 16 + qy * 2;
        const int c0 = min(x, W - 1), c1 = min(x + 1, W - 1);
        const uint32_t p00 = jpeg_raw_to_rgb<FMT>(px[it][0]), p01 = jpeg_raw_to_rgb<FMT>(px[it][1]);
        const uint32_t p10 = jpeg_raw_to_rgb<FMT>(px[it][2]), p11 = jpeg_raw_to_rgb<FMT>(px[it][3]);
        const int bias = (qx & 1) ? 2 : 1; /* chroma column parity within the MCU == global parity */
        const int cb = (jpeg_cb(p00) + jpeg_cb(p01) + jpeg_cb(p10) + jpeg_cb(p11) + bias) >> 2;
        const int cr = (jpeg_cr(p00) + jpeg_cr(p01) + jpeg_cr(p10) + jpeg_cr(p11) + bias) >> 2;
        cbtile[qy * CT_STRIDE + slot * 8 + qx] = (uint8_t)cb;
        crtile[qy * CT_STRIDE + slot * 8 + qx] = (uint8_t)cr;
        /* luma: plain clamping of both coordinates */
        uint32_t q00 = p00, q01 = p01, q10 = p10, q11 = p11;
        if (y + 1 >= H) {
            const int y0 = min(y, H - 1), y1 = min(y + 1, H - 1);
            q00 = jpeg_fetch_rgb<FMT>(job, c0, y0);
            q01 = jpeg_fetch_rgb<FMT>(job, c1, y0);
            q10 = jpeg_fetch_rgb<FMT>(job, c0, y1);
            q11 = jpeg_fetch_rgb<FMT>(job, c1, y1);
        }
        uint8_t *yp = &ytile[(qy * 2) * YT_STRIDE + slot * 16 + qx * 2];
        yp[0] = (uint8_t)jpeg_y(q00);
        yp[1] = (uint8_t)jpeg_y(q01);
        yp[YT_STRIDE] = (uint8_t)jpeg_y(q10);
        yp[YT_STRIDE + 1] = (uint8_t)jpeg_y(q11);
    }
}

/* One 2x2 quad the slow way, every coordinate clamped: the image's right and
 * bottom edges, where MCUs hang over (same rules as jpeg_load_tiles). */
template <int FMT>
__device__ __forceinline__ void jpeg_quad_edge(const JpegDevJob &job, uint8_t *ytile, uint8_t *cbtile,
                                               uint8_t *crtile, int slot, int qx, int qy, int mcx, int mcy)
{
    const int W = job.width, H = job.height;
    const int ch = (H + 1) >> 1;
    const int x = mcx * 16 + qx * 2, y = mcy * 16 + qy * 2;
    const int cyc = min(mcy * 8 + qy, ch - 1);
    const int r0 = min(2 * cyc, H - 1), r1 = min(2 * cyc + 1, H - 1);
    const int c0 = min(x, W - 1), c1 = min(x + 1, W - 1);
    const uint32_t p00 = jpeg_fetch_rgb<FMT>(job, c0, r0), p01 = jpeg_fetch_rgb<FMT>(job, c1, r0);
    const uint32_t p10 = jpeg_fetch_rgb<FMT>(job, c0, r1), p11 = jpeg_fetch_rgb<FMT>(job, c1, r1);
    const int bias = (qx & 1) ? 2 : 1;
    cbtile[qy * CT_STRIDE + slot * 8 + qx] =
        (uint8_t)((jpeg_cb(p00) + jpeg_cb(p01) + jpeg_cb(p10) + jpeg_cb(p11) + bias) >> 2);
    crtile[qy * CT_STRIDE + slot * 8 + qx] =
        (uint8_t)((jpeg_cr(p00) + jpeg_cr(p01) + jpeg_cr(p10) + jpeg_cr(p11) + bias) >> 2);
    const int y0 = min(y, H - 1), y1 = min(y + 1, H - 1);
    uint8_t *yp = &ytile[(qy * 2) * YT_STRIDE + slot * 16 + qx * 2];
    yp[0] = (uint8_t)jpeg_y(jpeg_fetch_rgb<FMT>(job, c0, y0));
    yp[1] = (uint8_t)jpeg_y(jpeg_fetch_rgb<FMT>(job, c1, y0));
    yp[YT_STRIDE] = (uint8_t)jpeg_y(jpeg_fetch_rgb<FMT>(job, c0, y1));
    yp[YT_STRIDE + 1] = (uint8_t)jpeg_y(jpeg_fetch_rgb<FMT>(job, c1, y1));
}

/* libjpeg jccolor.c on a pixel stored b,g,r,pad */
__device__ __forceinline__ void jpeg_ycc_bgrx(uint32_t p, int &yy, int &cb, int &cr)
{
    const int b = p & 0xff, g = (p >> 8) & 0xff, r = (p >> 16) & 0xff;
    yy = (19595 * r + 38470 * g + 7471 * b + 32768) >> 16;
    cb = (-11059 * r - 21709 * g + (b << 15) + ((128 << 16) + 32767)) >> 16;
    cr = ((r << 15) - 27439 * g - 5329 * b + ((128 << 16) + 32767)) >> 16;
}

/* Two pixels stored b,g,r in their low three bytes at once.  The same integers as
 * jpeg_ycc_bgrx, computed in single precision: every product and partial sum is a
 * multiple of 2^-16 below 2^8 (at most 2^24 - 1 in units of 2^-16: K + 32768 x 255 =
 * 16 777 215), so nothing is rounded, and the conversion back truncates exactly where
 * the shift by 16 does.  A byte becomes a float in one instruction, which pays for the
 * byte extraction, and two pixels share every multiply-add (v_pk_fma_f32). */
__device__ __forceinline__ void jpeg_ycc_pair(uint32_t p0, uint32_t p1, int &y0, int &y1, int &cb0,
                                              int &cb1, int &cr0, int &cr1)
{
    const float s = 1.0f / 65536.0f, k = (float)((128 << 16) + 32767) * s;
    typedef spice_f2 f2;
    const f2 b = {(float)(p0 & 0xff), (float)(p1 & 0xff)};
    const f2 g = {(float)((p0 >> 8) & 0xff), (float)((p1 >> 8) & 0xff)};
    const f2 r = {(float)((p0 >> 16) & 0xff), (float)((p1 >> 16) & 0xff)};
    f2 y = spice_fma2(r, (f2)(19595.0f * s), (f2)(32768.0f * s));
    y = spice_fma2(g, (f2)(38470.0f * s), y);
    y = spice_fma2(b, (f2)(7471.0f * s), y);
    f2 cb = spice_fma2(b, (f2)(0.5f), (f2)(k));
    cb = spice_fma2(r, (f2)(-11059.0f * s), cb);
    cb = spice_fma2(g, (f2)(-21709.0f * s), cb);
    f2 cr = spice_fma2(r, (f2)(0.5f), (f2)(k));
    cr = spice_fma2(g, (f2)(-27439.0f * s), cr);
    cr = spice_fma2(b, (f2)(-5329.0f * s), cr);
    y0 = (int)y.x;
    y1 = (int)y.y;
    cb0 = (int)cb.x;
    cb1 = (int)cb.y;
    cr0 = (int)cr.x;
    cr1 = (int)cr.y;
}

/* Four stored pixels from the words of a row piece (4, 3 or 2 words by format) */
template <int FMT>
__device__ __forceinline__ void jpeg_unpack4(const uint32_t *w, uint32_t *raw)
{
    if (FMT == 8 || FMT == 9) {
        raw[0] = w[0];
        raw[1] = w[1];
        raw[2] = w[2];
        raw[3] = w[3];
    } else if (FMT == 7 || FMT == JPEG_FMT_RGB24) {
        raw[0] = w[0] & 0xffffffu;
        raw[1] = (w[0] >> 24) | ((w[1] & 0xffffu) << 8);
        raw[2] = (w[1] >> 16) | ((w[2] & 0xffu) << 16);
        raw[3] = w[2] >> 8;
    } else {
        raw[0] = w[0] & 0xffffu;
        raw[1] = w[0] >> 16;
        raw[2] = w[1] & 0xffffu;
        raw[3] = w[1] >> 16;
    }
}

template <int FMT>
__device__ __forceinline__ void jpeg_ycc(uint32_t raw, int &yy, int &cb, int &cr)
{
    if (FMT == 8 || FMT == 9) {
        jpeg_ycc_bgrx(raw, yy, cb, cr);
    } else {
        const uint32_t rgb = jpeg_raw_to_rgb<FMT>(raw);
        yy = jpeg_y(rgb);
        cb = jpeg_cb(rgb);
        cr = jpeg_cr(rgb);
    }
}

/* Phase 1, the wide way: a lane takes four pixels of two rows at a time, each
 * row piece (16, 12 or 8 bytes) in one request; all of a lane's requests are
 * out before the first pixel is converted.  Items that touch the image's right
 * or bottom edge go the clamped way, quad by quad.  A 16- or 24-bit image whose
 * rows do not start on 4-byte boundaries (a stream's sub-rectangle can start
 * anywhere) is loaded pixel by pixel instead. */
template <int FMT>
__device__ __forceinline__ void jpeg_load_tiles_wide(const JpegDevJob &job, uint8_t *ytile, uint8_t *cbtile,
                                                     uint8_t *crtile, int tid, int ns, bool has_halo, int hx,
                                                     int hy, int mx0, int my)
{
    constexpr int PXB = (FMT == 8 || FMT == 9) ? 4 : (FMT == 7 || FMT == JPEG_FMT_RGB24) ? 3 : 2;
    constexpr int NW = PXB; /* words in four pixels */
    const int W = job.width, H = job.height;
    if (PXB != 4 && (((uint64_t)(uintptr_t)job.pixels | (uint64_t)job.stride) & 3) != 0) {
        jpeg_load_tiles<FMT>(job, ytile, cbtile, crtile, tid, ns, has_halo, hx, hy, mx0, my);
        return;
    }
    const int cols = ns * 4; /* four-pixel columns of a tile row */
    const int64_t main_row = (int64_t)(my * 16) * job.stride, halo_row = (int64_t)(hy * 16) * job.stride;
    /* Items are (line pair qy of the MCU row, four-pixel column cx).  A lane keeps its
     * line pair and walks the columns 32 apart, so that the line's address, the tile
     * rows and the bounds in y are worked out once; only the columns from 160 on, at
     * most 4 x 8 items, are dealt out again, to the first 32 lanes.  Column cx lies
     * at x = 16 (mx0 - 1) + 4 cx, but for the halo's four (cx < 4), which belong to
     * the MCU coded before this group, maybe on the MCU row above. */
    constexpr int FULL = (JSLOTS * 4) / 32, ITERS = FULL + 1;
    static_assert(JT == 256 && (JSLOTS * 4 - FULL * 32) * 8 <= 32, "tail items fit the first 32 lanes");
    const int rq = tid >> 5, c0 = tid & 31;
    const int xb = (mx0 - 1) * 16;
    const JPEG_GLOBAL uint8_t *const pixels = (const JPEG_GLOBAL uint8_t *)(uintptr_t)job.pixels;
    uint32_t px[ITERS][2 * NW];
    unsigned fast = 0;
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const int qy = it < FULL ? rq : tid >> 2;
        const int cx = it < FULL ? c0 + 32 * it : FULL * 32 + (tid & 3);
#pragma unroll
        for (int k = 0; k < 2 * NW; k++) {
            px[it][k] = 0;
        }
        if (cx < cols && (it < FULL || tid < 32)) {
            const bool halo = it == 0 && cx < 4;
            const int x = (halo ? hx * 16 : xb) + cx * 4, y = (halo ? hy : my) * 16 + qy * 2;
            if ((!halo || has_halo) && x + 3 < W && y + 1 < H) {
                /* the MCU row's first line is the same for every lane (scalar 64-bit
                 * product); the line inside it and the column fit 24-bit multiplies */
                const JPEG_GLOBAL uint8_t *row =
                    pixels + (halo ? halo_row : main_row) + __mul24(qy * 2, job.stride) + x * PXB;
                const JPEG_GLOBAL uint32_t *a = (const JPEG_GLOBAL uint32_t *)row;
                const JPEG_GLOBAL uint32_t *b = (const JPEG_GLOBAL uint32_t *)(row + job.stride);
#pragma unroll
                for (int k = 0; k < NW; k++) {
                    px[it][k] = a[k];
                    px[it][NW + k] = b[k];
                }
                fast |= 1u << it;
            }
        }
    }
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        const int qy = it < FULL ? rq : tid >> 2;
        const int cx = it < FULL ? c0 + 32 * it : FULL * 32 + (tid & 3);
        if (cx >= cols || (it >= FULL && tid >= 32)) {
            continue;
        }
        const bool halo = it == 0 && cx < 4;
        if (halo && !has_halo) {
            continue;
        }
        if (fast & (1u << it)) {
            uint32_t raw[8];
            jpeg_unpack4<FMT>(&px[it][0], &raw[0]);
            jpeg_unpack4<FMT>(&px[it][NW], &raw[4]);
            int yy[8], cb[8], cr[8];
            if (FMT == 8 || FMT == 9 || FMT == 7) {
#pragma unroll
                for (int k = 0; k < 8; k += 2) {
                    jpeg_ycc_pair(raw[k], raw[k + 1], yy[k], yy[k + 1], cb[k], cb[k + 1], cr[k], cr[k + 1]);
                }
            } else {
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    jpeg_ycc<FMT>(raw[k], yy[k], cb[k], cr[k]);
                }
            }
            /* slot * 16 + (cx & 3) * 4 is cx * 4: a tile row is the columns side by side */
            uint8_t *yp = &ytile[__mul24(qy * 2, YT_STRIDE) + cx * 4];
            *(uint32_t *)yp = (uint32_t)yy[0] | ((uint32_t)yy[1] << 8) | ((uint32_t)yy[2] << 16) | ((uint32_t)yy[3] << 24);
            *(uint32_t *)(yp + YT_STRIDE) =
                (uint32_t)yy[4] | ((uint32_t)yy[5] << 8) | ((uint32_t)yy[6] << 16) | ((uint32_t)yy[7] << 24);
            /* chroma column parity inside the MCU equals the global parity: bias 1, then 2 */
            const uint32_t cb0 = (uint32_t)(cb[0] + cb[1] + cb[4] + cb[5] + 1) >> 2;
            const uint32_t cb1 = (uint32_t)(cb[2] + cb[3] + cb[6] + cb[7] + 2) >> 2;
            const uint32_t cr0 = (uint32_t)(cr[0] + cr[1] + cr[4] + cr[5] + 1) >> 2;
            const uint32_t cr1 = (uint32_t)(cr[2] + cr[3] + cr[6] + cr[7] + 2) >> 2;
            const int co = __mul24(qy, CT_STRIDE) + cx * 2;
            *(uint16_t *)&cbtile[co] = (uint16_t)(cb0 | (cb1 << 8));
            *(uint16_t *)&crtile[co] = (uint16_t)(cr0 | (cr1 << 8));
        } else {
            const int slot = cx >> 2;
            const int mcx = slot == 0 ? hx : mx0 + slot - 1;
            const int mcy = slot == 0 ? hy : my;
            jpeg_quad_edge<FMT>(job, ytile, cbtile, crtile, slot, (cx & 3) * 2, qy, mcx, mcy);
            jpeg_quad_edge<FMT>(job, ytile, cbtile, crtile, slot, (cx & 3) * 2 + 1, qy, mcx, mcy);
        }
    }
}

/* Always inlined.  One development build, in which the compiler had emitted this body
 * as a call, took a memory access fault on the GPU; forcing the inline removed it.  The
 * first explanation -- the callee not being held to the kernel's register limit
 * (JPEG_FOUR_PER_CU) -- does not stand: rebuilt as a call (noinline, hipcc of ROCm 7.2,
 * gfx950) the callee stays within v0..v127, because a function without attributes is
 * compiled for work-groups of up to 1024 lanes = four waves per SIMD, this kernel's own
 * budget, and the kernel's descriptor covers the callee (.vgpr_count 128,
 * .private_segment_fixed_size 256 for the callee-saved v40..v127, LDS 38 936 B).  A
 * callee over budget would lower the occupancy, not fault.  The faulting source itself
 * (an intermediate state of the Huffman walk) was never committed, so its fault is not
 * explained; what is shipped has no call in either strip kernel. */
template <bool IFAST>
__device__ __forceinline__ void jpeg_strip_body(const JpegDevJob &job, const JpegDevJob &full, const uint32_t *huff_g, int group)
{
    /* the pixel tiles are dead once every lane holds its block in registers, so the
     * coefficients take their place: with the tables, the quantiser's constants and the bit rings 40 760 B per workgroup, four workgroups per CU */
    constexpr int kTileBytes = 16 * YT_STRIDE + 2 * 8 * CT_STRIDE;
    /* one MCU's worth more than the group has: where the halo MCU's lanes put the
     * coefficients nobody reads (only its DC values are wanted) */
    constexpr int kCoefBytes = (JG_MCUS + 1) * 6 * JCOEF_STRIDE * 4;
    __shared__ __align__(16) uint8_t pool[kCoefBytes > kTileBytes ? kCoefBytes : kTileBytes];
    uint8_t *const ytile = pool;
    uint8_t *const cbtile = pool + 16 * YT_STRIDE;
    uint8_t *const crtile = cbtile + 8 * CT_STRIDE;
    uint32_t *const coef = (uint32_t *)pool;
    __shared__ int dcq[JSLOTS * 6];
    __shared__ uint32_t huff[JHUFF_WORDS];
    __shared__ uint32_t rings[(JT / 64) * JRING_WORDS];
    /* the quantiser's constants as the record holds them: rcp_y, rcp_c, qhalf_y, qhalf_c */
    __shared__ __align__(16) float qtab[4 * 64];
    static_assert(JT == 4 * 64, "a lane stages one of the quantiser's 256 constants");
    static_assert(offsetof(JpegDevJob, rcp_c) == offsetof(JpegDevJob, rcp_y) + 256 &&
                      offsetof(JpegDevJob, qhalf_y) == offsetof(JpegDevJob, rcp_y) + 512 &&
                      offsetof(JpegDevJob, qhalf_c) == offsetof(JpegDevJob, rcp_y) + 768,
                  "the four tables lie side by side");

    const int tid = (int)threadIdx.x;
    /* Where the group lies is the same for every lane, but the divisions are done
     * in vector registers: moved to scalar ones here, or the pixel loaders, which
     * use them for every item, fetch them back from scratch memory each time. */
    /* group / groups_per_row by the record's reciprocal: one scalar multiply instead of
     * a division worked out in vector registers at the head of every workgroup */
    const int my = job.groups_per_row == 1 ? group : (int)(((uint64_t)(uint32_t)group * job.groups_rcp) >> 32);
    const int first_strip = (group - my * job.groups_per_row) * JPEG_GROUP_STRIPS;
    const int mx0 = first_strip * JPEG_STRIP_MCUS;
    const int nm = min(JG_MCUS, job.mcu_cols - mx0);
    const bool has_halo = group != 0;
    const int ns = nm + 1;
    /* the MCU coded just before this strip: the one to the left, or the last of the row above */
    const int hy = !has_halo ? 0 : mx0 > 0 ? my : my - 1;
    const int hx = !has_halo ? 0 : mx0 > 0 ? mx0 - 1 : job.mcu_cols - 1;

    /* The Huffman tables are asked for here and put into LDS after the pixel phase: a
     * copy loop at this point is up to three global round trips one after the other
     * (load, wait, store) before the first pixel is even requested. */
    constexpr int kHuffTurns = (JHUFF_WORDS + JT - 1) / JT;
    uint32_t huff_w[kHuffTurns];
#pragma unroll
    for (int k = 0; k < kHuffTurns; k++) {
        const int i = tid + k * JT;
        huff_w[k] = i < JHUFF_WORDS ? huff_g[i] : 0u;
    }
    /* ... and so are the quantiser's constants, one per lane: read row by row through
     * the scalar unit in the middle of the quantiser, each row was a scalar round trip
     * the wavefront sat out */
    const float qtab_w = full.rcp_y[tid];
    for (int i = tid; i < (JT / 64) * JRING_WORDS; i += JT) {
        rings[i] = 0;
    }

    /* ---- phase 1 */
    switch (job.format) {
    case 8:
    case 9:
        jpeg_load_tiles_wide<8>(job, ytile, cbtile, crtile, tid, ns, has_halo, hx, hy, mx0, my);
        break;
    case 7:
        jpeg_load_tiles_wide<7>(job, ytile, cbtile, crtile, tid, ns, has_halo, hx, hy, mx0, my);
        break;
    case JPEG_FMT_RGB24:
        jpeg_load_tiles_wide<JPEG_FMT_RGB24>(job, ytile, cbtile, crtile, tid, ns, has_halo, hx, hy, mx0, my);
        break;
    default:
        jpeg_load_tiles_wide<6>(job, ytile, cbtile, crtile, tid, ns, has_halo, hx, hy, mx0, my);
        break;
    }
#pragma unroll
    for (int k = 0; k < kHuffTurns; k++) {
        const int i = tid + k * JT;
        if (i < JHUFF_WORDS) {
            huff[i] = huff_w[k];
        }
    }
    qtab[tid] = qtab_w;
    __syncthreads();

    /* ---- phase 2: one lane per block.  Luma lanes are ordered so that a wave
     * reads consecutive 8-byte columns of one tile row. */
    int c[64];
    jf2 pf[32];            /* the block as pairs of floats (IFAST) */
    int slot = 0, bidx = 0; /* block's MCU slot and its index 0..5 inside the MCU */
    bool have = false, real = false;
    const uint8_t *src = pool; /* the block's first row in its tile */
    int rstride = 0;
    if (tid < 4 * ns) {
        const int yi = tid / (2 * ns), col = tid - yi * 2 * ns;
        slot = col >> 1;
        const int xi = col & 1;
        bidx = yi * 2 + xi;
        have = true;
        const int mcx = slot == 0 ? hx : mx0 + slot - 1, mcy = slot == 0 ? hy : my;
        real = (mcy * 2 + yi) < job.yb_h && (mcx * 2 + xi) < job.yb_w;
        src = &ytile[(yi * 8) * YT_STRIDE + col * 8];
        rstride = YT_STRIDE;
    } else if (tid < 6 * ns) {
        const int k = tid - 4 * ns;
        const bool is_cr = k >= ns;
        slot = is_cr ? k - ns : k;
        bidx = is_cr ? 5 : 4;
        have = true;
        real = true;
        src = (is_cr ? crtile : cbtile) + slot * 8;
        rstride = CT_STRIDE;
    }
    if (have && real) {
        if (IFAST) {
            /* rows two at a time: a pair holds the same column of both */
#pragma unroll
            for (int rp = 0; rp < 4; rp++) {
                const uint2 v = *(const uint2 *)(src + (2 * rp) * rstride);
                const uint2 w = *(const uint2 *)(src + (2 * rp + 1) * rstride);
                pf[rp * 8 + 0] = jpeg_f2((float)(v.x & 0xff), (float)(w.x & 0xff));
                pf[rp * 8 + 1] = jpeg_f2((float)((v.x >> 8) & 0xff), (float)((w.x >> 8) & 0xff));
                pf[rp * 8 + 2] = jpeg_f2((float)((v.x >> 16) & 0xff), (float)((w.x >> 16) & 0xff));
                pf[rp * 8 + 3] = jpeg_f2((float)(v.x >> 24), (float)(w.x >> 24));
                pf[rp * 8 + 4] = jpeg_f2((float)(v.y & 0xff), (float)(w.y & 0xff));
                pf[rp * 8 + 5] = jpeg_f2((float)((v.y >> 8) & 0xff), (float)((w.y >> 8) & 0xff));
                pf[rp * 8 + 6] = jpeg_f2((float)((v.y >> 16) & 0xff), (float)((w.y >> 16) & 0xff));
                pf[rp * 8 + 7] = jpeg_f2((float)(v.y >> 24), (float)(w.y >> 24));
            }
        } else {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const uint2 v = *(const uint2 *)(src + r * rstride);
                c[r * 8 + 0] = (int)(v.x & 0xff) - 128;
                c[r * 8 + 1] = (int)((v.x >> 8) & 0xff) - 128;
                c[r * 8 + 2] = (int)((v.x >> 16) & 0xff) - 128;
                c[r * 8 + 3] = (int)(v.x >> 24) - 128;
                c[r * 8 + 4] = (int)(v.y & 0xff) - 128;
                c[r * 8 + 5] = (int)((v.y >> 8) & 0xff) - 128;
                c[r * 8 + 6] = (int)((v.y >> 16) & 0xff) - 128;
                c[r * 8 + 7] = (int)(v.y >> 24) - 128;
            }
        }
    }
    if (slot == 0 && !has_halo) {
        have = false;
    }
    __syncthreads(); /* last read of the tiles; the coefficients overwrite them */
    if (have && real) {
        if (IFAST) {
            jpeg_fdct_ifast_f2(pf);
        } else {
            jpeg_fdct<false>(c);
            /* the quantiser below works on pairs of floats (|coefficient| < 2^16: exact) */
#pragma unroll
            for (int j = 0; j < 32; j++) {
                pf[j] = jpeg_f2((float)c[2 * j], (float)c[2 * j + 1]);
            }
        }
    }
    /* jcdctmgr.c's sign(v) x floor((|v| + d/2) / d) on the pairs the DCT left:
     * a = v + copysign(d/2, v); a x rcp, with rcp the float at or just above
     * 1/d, lies at or above a/d and below the next integer (|a| < 2^18), so
     * truncating it is the division; two coefficients share the add and the
     * multiply.  (Skipping the division where a whole wavefront's coefficients
     * round to zero, as the integer quantiser did, no longer pays: the votes cost
     * what the six instructions of a pair do.)  The constants come from LDS a row of
     * eight at a time, the same addresses for all luma lanes and for all chroma lanes. */
    if (have && real) {
        /* everything but the sign bit, in a vector register: the sign transfer
         * below is then one v_bfi_b32 */
        uint32_t magn = 0x7fffffffu;
        SPICE_OPAQUE_VGPR(magn);
        /* The quotients go straight to the block's place in LDS, 16 bits each at
         * its zigzag position, as they are made: none of them stays in a
         * register waiting for its neighbour in zigzag order. */
        constexpr int izz[64] = {JIZZ_LIST};
        uint16_t *dst16 = (uint16_t *)(coef + ((slot > 0 ? slot - 1 : JG_MCUS) * 6 + bidx) * JCOEF_STRIDE);
        const jf8 *rc = (const jf8 *)&qtab[bidx < 4 ? 0 : 64];
        const jf8 *hf = (const jf8 *)&qtab[bidx < 4 ? 128 : 192];
#pragma unroll
        for (int g = 0; g < 8; g++) {
            const jf8 hf8 = hf[g], rc8 = rc[g];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int j = g * 4 + u;
                const jf2 t = pf[j];
                /* sign of t over d/2: t ^ ((h ^ t) & magn), one v_bfi_b32 */
                const uint32_t b0 = jpeg_fbits(t[0]), b1 = jpeg_fbits(t[1]);
                const uint32_t s0 = b0 ^ ((jpeg_fbits(hf8[2 * u]) ^ b0) & magn);
                const uint32_t s1 = b1 ^ ((jpeg_fbits(hf8[2 * u + 1]) ^ b1) & magn);
                const jf2 a = t + jpeg_f2(jpeg_bitsf(s0), jpeg_bitsf(s1));
                const jf2 q = a * jpeg_f2(rc8[2 * u], rc8[2 * u + 1]);
                const int q0 = (int)q[0], q1 = (int)q[1];
                dst16[izz[2 * j]] = (uint16_t)q0;
                dst16[izz[2 * j + 1]] = (uint16_t)q1;
                if (j == 0) {
                    c[0] = q0;
                }
            }
        }
    }
    if (have && real) {
        dcq[slot * 6 + bidx] = c[0];
    }
    /* a dummy block is all zeros (its DC comes out of dcq below) */
    if (have && !real && slot > 0) {
        uint32_t *dst = coef + ((slot - 1) * 6 + bidx) * JCOEF_STRIDE;
#pragma unroll
        for (int j = 0; j < 32; j++) {
            dst[j] = 0;
        }
    }
    __syncthreads();
    /* dummy edge blocks repeat the DC of the block coded before them (jccoefct.c) */
    if (tid < ns && (tid > 0 || has_halo)) {
        const int mcx = tid == 0 ? hx : mx0 + tid - 1, mcy = tid == 0 ? hy : my;
        for (int b = 1; b < 4; b++) {
            const bool rl = (mcy * 2 + (b >> 1)) < job.yb_h && (mcx * 2 + (b & 1)) < job.yb_w;
            if (!rl) {
                dcq[tid * 6 + b] = dcq[tid * 6 + b - 1];
            }
        }
    }
    __syncthreads();
    /* DC differences replace the DC values in the coefficient store (jchuff.c
     * encode_one_block codes the difference to the previous block of the same
     * component; the four luma blocks of an MCU chain through each other) */
    if (tid < nm * 6) {
        const int bslot = tid / 6 + 1, bi = tid - (bslot - 1) * 6;
        int pred;
        if (bi == 0) {
            pred = (bslot == 1 && !has_halo) ? 0 : dcq[(bslot - 1) * 6 + 3];
        } else if (bi < 4) {
            pred = dcq[bslot * 6 + bi - 1];
        } else {
            pred = (bslot == 1 && !has_halo) ? 0 : dcq[(bslot - 1) * 6 + bi];
        }
        ((uint16_t *)coef)[tid * (JCOEF_STRIDE * 2)] = (uint16_t)(dcq[bslot * 6 + bi] - pred);
    }
    __syncthreads();

    /* ---- phase 3: Huffman, one wavefront per strip */
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63; /* scalar: the loop below is uniform */
    const int strip_in_row = first_strip + wave;
    if (strip_in_row < job.strips_per_row) {
        const int strip = my * job.strips_per_row + strip_in_row;
        const int m0 = wave * JPEG_STRIP_MCUS, m1 = min(m0 + JPEG_STRIP_MCUS, nm);
        uint32_t *ring = rings + wave * JRING_WORDS;
        JPEG_GLOBAL uint32_t *out_words =
            (JPEG_GLOBAL uint32_t *)(uintptr_t)(full.strip_buf + (size_t)strip * full.strip_stride);
        const uint16_t *coef16 = (const uint16_t *)coef;
        uint32_t bits = 0, flushed = 0; /* stream bits so far; words already in the slot */

        const uint32_t zrl = huff[0xf0], zrl_c = huff[256 + 0xf0];

        /* The whole strip at once, a lane per block: each lane walks its block's
         * coefficients in zigzag order, builds the block's code string in a
         * 64-bit register and learns its length; a scan over the lanes places
         * the strings.  Taken when the strip's bits fit the ring (video content
         * does, at a few tens of bits per block); blocks longer than 64 bits are
         * walked a second time and written code by code. */
        const int nblk = (m1 - m0) * 6;
        bool strip_done = false;
        {
            const bool on = lane < nblk;
            const int bi = lane % 6;
            const int ac_off = bi < 4 ? 0 : 256, dc_off = bi < 4 ? 512 : 528;
            const uint32_t z = bi < 4 ? zrl : zrl_c;
            const int zl = (int)(z >> 16);
            const uint32_t *cw = coef + (m0 * 6 + (on ? lane : 0)) * JCOEF_STRIDE;
            int len = 0;
            uint64_t acc = 0;
            int run = 0;
            if (on) {
                int kprev = 0; /* zigzag index of the last coefficient coded */
                /* the block's 32 words are asked for at once: read one by one, inside the
                 * tests below, each would cost the walk a whole LDS round trip */
                uint32_t xw[32];
#pragma unroll
                for (int w = 0; w < 32; w++) {
                    xw[w] = cw[w];
                }
#pragma unroll
                for (int w = 0; w < 32; w++) {
                    const uint32_t x = xw[w];
                    if (w > 0 && x == 0) {
                        continue;
                    }
#pragma unroll
                    for (int h = 0; h < 2; h++) {
                        const int k = 2 * w + h;
                        const int v = h ? (int)x >> 16 : (int)(int16_t)x;
                        if (k > 0 && v == 0) {
                            continue;
                        }
                        const int t = v < 0 ? -v : v;
                        const int nb = 32 - __clz(t);
                        const uint32_t vbits = jpeg_low_bits((uint32_t)(v + (v >> 31)), nb);
                        if (k == 0) {
                            const uint32_t e = huff[dc_off + nb];
                            len = (int)(e >> 16) + nb;
                            acc = ((e & 0xffff) << nb) | vbits;
                        } else {
                            const int r = k - 1 - kprev;
                            const uint32_t e = huff[ac_off + ((r & 15) << 4) + nb];
                            const int el = (int)(e >> 16) + nb;
                            const int nz16 = r >> 4;
                            for (int i = 0; i < nz16; i++) {
                                acc = (acc << zl) | (z & 0xffff);
                            }
                            acc = (acc << el) | (((e & 0xffff) << nb) | vbits);
                            len += nz16 * zl + el;
                            kprev = k;
                        }
                    }
                }
                if (kprev != 63) {
                    const uint32_t eob = huff[ac_off];
                    acc = (acc << (eob >> 16)) | (eob & 0xffff);
                    len += (int)(eob >> 16);
                }
            }
            const uint32_t inc = wave_scan_inclusive((uint32_t)len, lane);
            const uint32_t total = wave_broadcast(inc, 63);
            if (total <= JRING_WORDS * 32) {
                const uint32_t pos0 = inc - (uint32_t)len;
                if (len > 0 && len <= 64) {
                    if (len > 32) {
                        jpeg_ring_put(ring, pos0, (uint32_t)(acc >> 32), len - 32);
                        jpeg_ring_put(ring, pos0 + (uint32_t)(len - 32), (uint32_t)acc, 32);
                    } else {
                        jpeg_ring_put(ring, pos0, (uint32_t)acc, len);
                    }
                }
                if (__ballot(len > 64) != 0) {
                    if (len > 64) {
                        uint32_t pos = pos0;
                        run = 0;
                        for (int k = 0; k < 64; k++) {
                            const int v = (int)(int16_t)coef16[(m0 * 6 + lane) * (JCOEF_STRIDE * 2) + k];
                            const int t = v < 0 ? -v : v;
                            const int nb = 32 - __clz(t);
                            const uint32_t vbits = jpeg_low_bits((uint32_t)(v + (v >> 31)), nb);
                            if (k == 0) {
                                const uint32_t e = huff[dc_off + nb];
                                jpeg_ring_put(ring, pos, ((e & 0xffff) << nb) | vbits, (int)(e >> 16) + nb);
                                pos += (e >> 16) + (uint32_t)nb;
                            } else if (v != 0) {
                                for (int i = run >> 4; i > 0; i--) {
                                    jpeg_ring_put(ring, pos, z & 0xffff, zl);
                                    pos += (uint32_t)zl;
                                }
                                const uint32_t e = huff[ac_off + ((run & 15) << 4) + nb];
                                jpeg_ring_put(ring, pos, ((e & 0xffff) << nb) | vbits, (int)(e >> 16) + nb);
                                pos += (e >> 16) + (uint32_t)nb;
                                run = 0;
                            } else {
                                run++;
                            }
                        }
                        if (run) {
                            const uint32_t eob = huff[ac_off];
                            jpeg_ring_put(ring, pos, eob & 0xffff, (int)(eob >> 16));
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
                for (uint32_t i = (uint32_t)lane; i < (total + 31) >> 5; i += 64) {
                    const uint32_t w = ring[i];
                    ring[i] = 0;
                    out_words[i] = __builtin_bswap32(w);
                }
                __builtin_amdgcn_wave_barrier();
                bits = total;
                strip_done = true;
            }
        }

        int vnext = m0 < m1 && !strip_done ? (int)(int16_t)coef16[m0 * 6 * (JCOEF_STRIDE * 2) + lane] : 0;
        for (int b = m0 * 6; b < m1 * 6 && !strip_done; b++) {
            const int bi = b % 6;
            const int v = vnext;
            if (b + 1 < m1 * 6) {
                vnext = (int)(int16_t)coef16[(b + 1) * (JCOEF_STRIDE * 2) + lane];
            }
            const int ac_off = bi < 4 ? 0 : 256, dc_off = bi < 4 ? 512 : 528;

            /* one code per lane: DC (lane 0), a non-zero AC coefficient with the
             * zero run before it, or the end-of-block code when the block ends
             * in zeros (lane 63) */
            const uint64_t nz = __ballot(v != 0) & ~1ull;
            const int t = v < 0 ? -v : v, t2 = v + (v >> 31);
            const int nb = 32 - __clz(t);
            const uint32_t vbits = jpeg_low_bits((uint32_t)t2, nb);
            uint32_t total;
            if (nz == 0) {
                /* no AC coefficient at all (about every other block of video content):
                 * the DC code and the end-of-block code, both written by lane 0 */
                const uint32_t e = huff[dc_off + nb];
                const uint32_t eob = huff[ac_off];
                const int el = (int)(eob >> 16);
                const int len = (int)(e >> 16) + nb + el;
                if (lane == 0) {
                    jpeg_ring_put(ring, bits, ((((e & 0xffff) << nb) | vbits) << el) | (eob & 0xffff), len);
                }
                total = (uint32_t)__builtin_amdgcn_readfirstlane(len);
            } else {
                const uint64_t before = nz & ((1ull << lane) - 1);
                const int run = lane - 1 - (before ? 63 - __clzll((long long)before) : 0);
                const bool ac = v != 0 && lane != 0;
                const int idx = lane == 0 ? dc_off + nb : ac_off + (ac ? ((run & 15) << 4) + nb : 0);
                const uint32_t e = huff[idx];
                const int es = (int)(e >> 16);
                const uint32_t code = ((e & 0xffff) << nb) | vbits;
                int len = (ac || lane == 0 || lane == 63) ? es + nb : 0;
                /* runs of 16 zeros and more take ZRL codes first; rare */
                const int nzrl = ac ? run >> 4 : 0;
                const uint32_t z = bi < 4 ? zrl : zrl_c;
                const bool any_zrl = __ballot(nzrl != 0) != 0;
                if (any_zrl) {
                    len += nzrl * (int)(z >> 16);
                }
                const uint32_t inc = wave_scan_inclusive((uint32_t)len, lane);
                total = wave_broadcast(inc, 63);
                if (len) {
                    uint32_t pos = bits + inc - (uint32_t)len;
                    if (any_zrl) {
                        for (int i = 0; i < nzrl; i++) {
                            jpeg_ring_put(ring, pos, z & 0xffff, (int)(z >> 16));
                            pos += z >> 16;
                        }
                    }
                    jpeg_ring_put(ring, pos, code, es + nb);
                }
            }
            bits += total;
            __builtin_amdgcn_wave_barrier();
            /* complete words leave the ring: the stream is MSB first, memory wants bytes in order */
            const uint32_t done = bits >> 5;
            if (done - flushed >= 32 || b + 1 == m1 * 6) {
                const uint32_t upto = b + 1 == m1 * 6 ? (bits + 31) >> 5 : done;
                for (uint32_t i = flushed + (uint32_t)lane; i < upto; i += 64) {
                    const uint32_t w = ring[i & (JRING_WORDS - 1)];
                    ring[i & (JRING_WORDS - 1)] = 0;
                    out_words[i] = __builtin_bswap32(w);
                }
                flushed = upto;
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (lane == 0) {
            full.strip_bits[strip] = bits;
        }
    }
}

/* Four workgroups per CU: at most 128 VGPRs (106 / 112 used, nothing spilled) and 40 760 B of LDS each. */
template <bool IFAST>
__global__ void __launch_bounds__(JT) JPEG_FOUR_PER_CU
jpeg_strip_kernel(const JpegDevJob *jobs, const uint32_t *huff)
{
    const JpegDevJob &full = jobs[blockIdx.y];
    /* The image's geometry, sixteen words at the record's head, is copied here in one
     * go (a few wide scalar loads and one wait) and the loaders and the set-up work on
     * the copy: read field by field where each is first used, they were seven scalar
     * round trips one behind the other before the first pixel request.  Only these
     * members of the copy are ever set or read; the tables and the scratch pointers
     * stay behind `full`. */
    JpegDevJob job;
    {
        /* read as constant memory: the record is not written while the kernel runs, and
         * the compiler may then use scalar loads whatever stands in front of them */
        const JPEG_CONST JpegDevJob &head = *(const JPEG_CONST JpegDevJob *)(uintptr_t)&full;
        job.pixels = head.pixels;
        job.stride = head.stride;
        job.width = head.width;
        job.height = head.height;
        job.format = head.format;
        job.dct = head.dct;
        job.mcu_cols = head.mcu_cols;
        job.mcu_rows = head.mcu_rows;
        job.strips_per_row = head.strips_per_row;
        job.nstrips = head.nstrips;
        job.yb_w = head.yb_w;
        job.yb_h = head.yb_h;
        job.groups_per_row = head.groups_per_row;
        job.groups_rcp = head.groups_rcp;
    }
    const int group = (int)blockIdx.x;
    if (group >= job.groups_per_row * job.mcu_rows || (job.dct != 0) != IFAST) {
        return;
    }
    jpeg_strip_body<IFAST>(job, full, huff, group);
}

/* Exclusive scan of strip_bits into strip_bitoff (64-bit), one workgroup per job:
 * every lane sums a run of consecutive strips, the runs' sums are scanned once
 * across the workgroup, and every lane walks its run again to write the offsets. */
__global__ void __launch_bounds__(JT)
jpeg_offsets_kernel(const JpegDevJob *jobs)
{
    __shared__ uint64_t wave_tot[JT / 64];
    const JpegDevJob &job = jobs[blockIdx.x];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (job.nstrips + JT - 1) / JT;
    const int i0 = min(tid * per, job.nstrips), i1 = min(i0 + per, job.nstrips);
    uint64_t mine = 0;
    for (int i = i0; i < i1; i++) {
        mine += job.strip_bits[i];
    }
    uint64_t inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)inc, d);
        const uint32_t hi = (uint32_t)__shfl_up((int)(uint32_t)(inc >> 32), d);
        if (lane >= d) {
            inc += ((uint64_t)hi << 32) | lo;
        }
    }
    if (lane == 63) {
        wave_tot[wave] = inc;
    }
    __syncthreads();
    uint64_t base = 0, total = 0;
    for (int k = 0; k < JT / 64; k++) {
        const uint64_t t = wave_tot[k];
        if (k < wave) {
            base += t;
        }
        total += t;
    }
    uint64_t off = base + inc - mine;
    for (int i = i0; i < i1; i++) {
        job.strip_bitoff[i] = off;
        off += job.strip_bits[i];
    }
    if (tid == 0) {
        job.strip_bitoff[job.nstrips] = total;
    }
}

/* Unstuffed scan: byte j holds stream bits [8j, 8j+8); the tail is padded with
 * one-bits (jchuff.c flush_bits).  16 bytes per lane, 4 KiB per workgroup pass. */
__global__ void __launch_bounds__(JT)
jpeg_gather_kernel(const JpegDevJob *jobs)
{
    __shared__ int wave_sums[JT / 64];
    const JpegDevJob &job = jobs[blockIdx.y];
    const int tid = (int)threadIdx.x;
    const uint64_t tbits = job.strip_bitoff[job.nstrips];
    const uint64_t nbytes = (tbits + 7) >> 3;
    const uint64_t usable = nbytes < job.scan_cap ? nbytes : job.scan_cap;
    const uint32_t nchunks = (uint32_t)((usable + JPEG_CHUNK_BYTES - 1) / JPEG_CHUNK_BYTES);

    for (uint32_t chunk = blockIdx.x; chunk < nchunks && chunk < job.nchunks_cap; chunk += gridDim.x) {
        const uint64_t j0 = (uint64_t)chunk * JPEG_CHUNK_BYTES + (uint64_t)tid * 16;
        int ff = 0;
        if (j0 < usable) {
            uint64_t bit = j0 * 8;
            /* last strip whose offset is <= bit */
            int lo = 0, hi = job.nstrips - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (job.strip_bitoff[mid] <= bit) {
                    lo = mid;
                } else {
                    hi = mid - 1;
                }
            }
            int s = lo;
            uint64_t soff = job.strip_bitoff[s];
            uint32_t sbits = job.strip_bits[s];
            const uint8_t *sp = job.strip_buf + (size_t)s * job.strip_stride;
            /* The lane's 128 stream bits, most significant first in acc[0..3], a
             * strip's share at a time: five words of the strip's slot read as
             * big-endian numbers, aligned to the first wanted bit, cut to the
             * bits the strip still has, and ORed in behind what is there. */
            uint32_t acc[4] = {0, 0, 0, 0};
            int have = 0;
            const int want = (int)(nbytes - j0 < 16 ? nbytes - j0 : 16) * 8;
            while (have < want) {
                if (bit >= tbits) {
                    break; /* the rest is padding */
                }
                while (bit >= soff + sbits) {
                    s++;
                    soff = job.strip_bitoff[s];
                    sbits = job.strip_bits[s];
                    sp = job.strip_buf + (size_t)s * job.strip_stride;
                }
                const uint32_t local = (uint32_t)(bit - soff);
                const int left = (int)(sbits - local);
                const int take = left < want - have ? left : want - have;
                const uint32_t *wp = (const uint32_t *)(sp + ((local >> 3) & ~3u));
                const uint32_t t = local & 31;
                uint32_t v[4];
                {
                    uint32_t be[5];
#pragma unroll
                    for (int i = 0; i < 5; i++) {
                        be[i] = __builtin_bswap32(wp[i]);
                    }
#pragma unroll
                    for (int i = 0; i < 4; i++) {
                        v[i] = t ? (be[i] << t) | (be[i + 1] >> (32 - t)) : be[i];
                    }
                }
                /* keep the first `take` bits */
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int keep = take - 32 * i;
                    v[i] = keep >= 32 ? v[i] : keep <= 0 ? 0u : v[i] & ~(0xffffffffu >> keep);
                }
                /* shift right by `have` bits into place */
                const int ws = have >> 5, bs = have & 31;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int j = i - ws;
                    uint32_t x = 0;
                    if (j >= 0) {
                        x = v[j] >> bs;
                        if (bs && j >= 1) {
                            x |= v[j - 1] << (32 - bs);
                        }
                    }
                    acc[i] |= x;
                }
                have += take;
                bit += (uint64_t)take;
            }
            if (have < want) {
                /* pad with ones */
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int keep = have - 32 * i;
                    acc[i] |= keep >= 32 ? 0u : keep <= 0 ? 0xffffffffu : 0xffffffffu >> keep;
                }
            }
            uint32_t out4[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                /* bytes past the stream's end stay zero, as before */
                const int nb = want / 8 - 4 * i;
                const uint32_t vis = nb >= 4 ? acc[i] : nb <= 0 ? 0u : acc[i] & ~(0xffffffffu >> (8 * nb));
                /* bytes equal to 0xff: the zero bytes of ~vis among the nb real ones */
                const uint32_t n = ~vis;
                uint32_t z = ~(((n & 0x7f7f7f7fu) + 0x7f7f7f7fu) | n | 0x7f7f7f7fu);
                if (nb < 4) {
                    z = nb <= 0 ? 0u : z & ~(0xffffffffu >> (8 * nb));
                }
                ff += __popc(z);
                out4[i] = __builtin_bswap32(vis);
            }
            *(uint4 *)(job.scan_buf + j0) = make_uint4(out4[0], out4[1], out4[2], out4[3]);
        }
        int total = 0;
        jpeg_block_scan(ff, wave_sums, tid, &total);
        if (tid == 0) {
            job.chunk_ff[chunk] = (uint32_t)total;
        }
    }
}

/* Header, byte-stuffed scan (a zero after every 0xFF), EOI, result. */
__global__ void __launch_bounds__(JT)
jpeg_finish_kernel(const JpegDevJob *jobs)
{
    __shared__ int wave_sums[JT / 64];
    const JpegDevJob &job = jobs[blockIdx.y];
    const int tid = (int)threadIdx.x;
    const uint64_t tbits = job.strip_bitoff[job.nstrips];
    const uint64_t nbytes = (tbits + 7) >> 3;
    const bool fits_scan = nbytes <= job.scan_cap;
    const uint32_t nchunks = (uint32_t)((nbytes + JPEG_CHUNK_BYTES - 1) / JPEG_CHUNK_BYTES);
    const bool ok = fits_scan && nchunks <= job.nchunks_cap;

    if (blockIdx.x == 0) {
        for (uint32_t i = (uint32_t)tid; i < job.hdr_len && i < job.out_cap; i += JT) {
            job.out[i] = job.header[i];
        }
        int part = 0;
        if (ok) {
            for (uint32_t i = (uint32_t)tid; i < nchunks; i += JT) {
                part += (int)job.chunk_ff[i];
            }
        }
        int total_ff = 0;
        jpeg_block_scan(part, wave_sums, tid, &total_ff);
        if (tid == 0) {
            const uint64_t size = (uint64_t)job.hdr_len + nbytes + (uint64_t)total_ff + 2;
            if (!ok || size > job.out_cap) {
                *job.result = JPEG_DEV_ERR_NO_SPACE;
            } else {
                job.out[size - 2] = 0xff;
                job.out[size - 1] = 0xd9;
                *job.result = (int32_t)size;
            }
        }
    }
    if (!ok) {
        return;
    }
    for (uint32_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        int part = 0;
        for (uint32_t i = (uint32_t)tid; i < chunk; i += JT) {
            part += (int)job.chunk_ff[i];
        }
        int ff_before = 0;
        jpeg_block_scan(part, wave_sums, tid, &ff_before);

        const uint64_t j0 = (uint64_t)chunk * JPEG_CHUNK_BYTES + (uint64_t)tid * 16;
        uint32_t w4[4] = {0, 0, 0, 0};
        uint32_t ffw[4] = {0, 0, 0, 0}; /* 0x80 in every byte that is a real 0xff */
        int ff = 0, n = 0;
        if (j0 < nbytes) {
            const uint4 v = *(const uint4 *)(job.scan_buf + j0);
            w4[0] = v.x;
            w4[1] = v.y;
            w4[2] = v.z;
            w4[3] = v.w;
            n = (int)(nbytes - j0 < 16 ? nbytes - j0 : 16);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                /* bytes equal to 0xff among the real ones: the zero bytes of the complement */
                const int nb = n - 4 * i;
                const uint32_t c = ~w4[i];
                uint32_t z = ~(((c & 0x7f7f7f7fu) + 0x7f7f7f7fu) | c | 0x7f7f7f7fu);
                if (nb < 4) {
                    z = nb <= 0 ? 0u : z & (0xffffffffu >> (32 - 8 * nb));
                }
                ffw[i] = z;
                ff += __popc(z);
            }
        }
        int dummy = 0;
        const int ff_lane = jpeg_block_scan(ff, wave_sums, tid, &dummy);
        uint64_t pos = (uint64_t)job.hdr_len + j0 + (uint64_t)ff_before + (uint64_t)ff_lane;
        /* a word without 0xff goes out whole (the address need not be aligned), the
         * others and the tail byte by byte with their stuffing */
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int nb = n - 4 * i;
            if (nb >= 4 && ffw[i] == 0 && pos + 4 <= job.out_cap) {
                const uint32_t w = w4[i];
                __builtin_memcpy(job.out + pos, &w, 4);
                pos += 4;
                continue;
            }
            for (int k = 0; k < 4 && k < nb; k++) {
                const uint32_t b = (w4[i] >> (8 * k)) & 0xff;
                if (pos < job.out_cap) {
                    job.out[pos] = (uint8_t)b;
                }
                pos++;
                if (b == 0xff) {
                    if (pos < job.out_cap) {
                        job.out[pos] = 0;
                    }
                    pos++;
                }
            }
        }
    }
}

/* timed = false: no timing events round the launches (creating an event allocates) */
void jpeg_launch(const JpegDevJob *jobs, int njobs, int max_groups, bool any_islow, bool any_ifast,
                 const uint32_t *huff, hipStream_t stream, bool timed, hipEvent_t strips_after,
                 hipEvent_t strips_done)
{
    const dim3 grid((unsigned)max_groups, (unsigned)njobs);
    int t = -1;
    /* With two batches in flight on two streams, this batch's strip kernel starts when the
     * other's has ended (strips_after): the strip kernels, each of which fills the GPU, run
     * one after the other, and what overlaps them is the other batch's short bit-stream
     * kernels, its copies and the host's work.  (Left to overlap freely, two strip kernels
     * share the CUs: a step of 16 4K frames takes 0.221 ms instead of 0.245, but every
     * launch then lasts 0.33-0.40 ms and a kernel's duration says nothing any more.  All
     * strip kernels on one stream of their own, which spares them the cross-stream wait,
     * changes nothing: 0.244 against 0.244 ms per step.) */
    if (strips_after) {
        hipStreamWaitEvent(stream, strips_after, 0);
    }
    hipStream_t ks = stream;
    /* A timed strip kernel records its own begin and end (the dispatch's time stamps, what
     * a profiler reports): events of their own before and after it are packets the stream
     * works through one by one, about 5 us each, between this batch's strip kernel and the
     * other batch's, which waits for it. */
    hipEvent_t ta = nullptr, tb = nullptr;
    if (any_islow) {
        if (timed && spicegpu::timing_events("jpeg_strip_kernel_islow", &ta, &tb)) {
            SPICE_LAUNCH_TIMED(jpeg_strip_kernel<false>, grid, dim3(JT), ks, ta, tb, jobs, huff);
        } else {
            hipLaunchKernelGGL(jpeg_strip_kernel<false>, grid, dim3(JT), 0, ks, jobs, huff);
        }
    }
    if (any_ifast) {
        if (timed && spicegpu::timing_events("jpeg_strip_kernel_ifast", &ta, &tb)) {
            SPICE_LAUNCH_TIMED(jpeg_strip_kernel<true>, grid, dim3(JT), ks, ta, tb, jobs, huff);
        } else {
            hipLaunchKernelGGL(jpeg_strip_kernel<true>, grid, dim3(JT), 0, ks, jobs, huff);
        }
    }
    if (strips_done) {
        hipEventRecord(strips_done, stream);
    }
    /* the three bit-stream kernels share one pair of timing events: an event
     * between two launches costs about as much as the offsets kernel itself */
    t = timed ? spicegpu::timing_begin("jpeg_bitstream_kernels", stream) : -1;
    hipLaunchKernelGGL(jpeg_offsets_kernel, dim3((unsigned)njobs), dim3(JT), 0, stream, jobs);
    /* a workgroup takes the scan 4 KiB at a time; a group of strips (10 240 pixels)
     * leaves about that much at video qualities, so the largest job gets a workgroup
     * per few chunks and the passes are not a chain of dependent round trips (16 4K
     * frames: 0.048 -> 0.041 ms for the three kernels with 256 instead of 64) */
    unsigned per_job = (unsigned)max_groups / 3u;
    per_job = per_job < 16u ? 16u : (per_job > 256u ? 256u : per_job);
    hipLaunchKernelGGL(jpeg_gather_kernel, dim3(per_job, (unsigned)njobs), dim3(JT), 0, stream, jobs);
    hipLaunchKernelGGL(jpeg_finish_kernel, dim3(per_job, (unsigned)njobs), dim3(JT), 0, stream, jobs);
    spicegpu::timing_end(t, stream);
}

/* ---------------------------------------------------------------------------
 * One frame through the same kernels with no allocation of any kind: for callers
 * that cannot allocate device memory where they call from (tight serving loops,
 * harnesses that forbid it; the call waits for its result on the stream, so it is
 * not for code under HIP graph capture).  What
 * spice_gpu_jpeg_encode_batch takes from hipMalloc'ed, growing buffers -- the job
 * record, the Huffman tables, the strips' bit slots, their counts and offsets,
 * the unstuffed scan and the 0xFF counts -- lives in one array of the code
 * object instead.  Every GPU the library runs on holds its own copy of the array
 * until the process ends.
 *
 * Layout of the array (byte offsets):
 *   0        the JpegDevJob of the call in progress
 *   3072     its result (int32)
 *   4096     Huffman tables, JPEG_HUFF_WORDS words (written once, at set-up)
 *   8192     strip_bits      JPEG_NA_MAX_STRIPS words
 *   12288    strip_bitoff    JPEG_NA_MAX_STRIPS + 1 quad words
 *   24576    chunk_ff        words, one per 4 KiB of the unstuffed scan
 *   36864    scan_buf        JPEG_NA_MAX_OUT + 512 bytes
 *   ...      strip_buf       JPEG_NA_MAX_STRIPS slots of JPEG_NA_STRIP_STRIDE bytes
 * No kernel of the pipeline expects any of it to be zero: every strip writes its
 * count, offsets and chunk counts are written for every entry that is read, and
 * the gather kernel cuts what it reads of a slot to the bits the strip produced.
 */
#define JPEG_NA_MAX_STRIPS 1024          /* e.g. 1920 x 1088: 12 strips x 68 MCU rows = 816 */
#define JPEG_NA_MAX_OUT (4u << 20)       /* largest out_capacity */
#define JPEG_NA_STRIP_STRIDE (JPEG_MCU_SLOT_BYTES * JPEG_STRIP_MCUS)
#define JPEG_NA_OFF_RESULT 3072u
#define JPEG_NA_OFF_HUFF 4096u
#define JPEG_NA_OFF_BITS 8192u
#define JPEG_NA_OFF_BITOFF 12288u
#define JPEG_NA_OFF_CHUNKFF 24576u
#define JPEG_NA_OFF_SCAN 36864u
#define JPEG_NA_SCAN_BYTES (JPEG_NA_MAX_OUT + 512u)
#define JPEG_NA_OFF_STRIPS (JPEG_NA_OFF_SCAN + JPEG_NA_SCAN_BYTES)
#define JPEG_NA_BYTES (JPEG_NA_OFF_STRIPS + JPEG_NA_MAX_STRIPS * JPEG_NA_STRIP_STRIDE + 256u)
#define JPEG_NA_MAX_DEVICES 64

static_assert(sizeof(JpegDevJob) <= JPEG_NA_OFF_RESULT, "job record overlaps the result");
static_assert(JPEG_HUFF_WORDS * 4 <= JPEG_NA_OFF_BITS - JPEG_NA_OFF_HUFF, "Huffman tables overlap");
static_assert((JPEG_NA_MAX_STRIPS + 1) * 8 <= JPEG_NA_OFF_CHUNKFF - JPEG_NA_OFF_BITOFF, "offsets overlap");
static_assert((JPEG_NA_SCAN_BYTES / JPEG_CHUNK_BYTES + 2) * 4 <= JPEG_NA_OFF_SCAN - JPEG_NA_OFF_CHUNKFF,
              "chunk counts overlap");

/* In host code this name is a stand-in, not a device address: the address comes
 * from hipGetSymbolAddress, once per device (jpeg_noalloc_setup). */
static __device__ __align__(256) uint8_t jpeg_na_scratch[JPEG_NA_BYTES];

/* device address of jpeg_na_scratch on each GPU this process has initialised */
static uint8_t *g_jpeg_na_ptr[JPEG_NA_MAX_DEVICES];
/* a pinned host word per GPU, made at set-up, where a call's result lands: a copy into
 * pageable memory would go through a staging buffer of the runtime's */
static int32_t *g_jpeg_na_result[JPEG_NA_MAX_DEVICES];

bool jpeg_fill_job(JpegDevJob *j, const void *pixels, int format, int width, int height, long stride,
                   int quality, int dct, void *out, uint32_t out_cap);
void jpeg_huff_words(uint32_t *h);

/* the job record travels as a kernel argument: no host-to-device copy is needed */
__global__ void __launch_bounds__(64)
jpeg_na_put_job(const JpegDevJob job, JpegDevJob *dst)
{
    if (threadIdx.x == 0) {
        *dst = job;
    }
}

/* Called by spice_gpu_init with `device` current. */
bool jpeg_noalloc_setup(int device)
{
    if (device < 0 || device >= JPEG_NA_MAX_DEVICES) {
        spicegpu::set_error("device %d: the library handles devices 0..%d", device, JPEG_NA_MAX_DEVICES - 1);
        return false;
    }
    if (g_jpeg_na_ptr[device]) {
        return true;
    }
    uint8_t *ptr = nullptr;
    if (!spicegpu::check(hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(jpeg_na_scratch)),
                         "hipGetSymbolAddress(jpeg_na_scratch)") || !ptr) {
        return false;
    }
    uint32_t h[JPEG_HUFF_WORDS];
    jpeg_huff_words(h);
    if (!spicegpu::check(hipMemcpy(ptr + JPEG_NA_OFF_HUFF, h, sizeof(h), hipMemcpyHostToDevice),
                         "upload Huffman tables")) {
        return false;
    }
    if (!g_jpeg_na_result[device] &&
        !spicegpu::check(hipHostMalloc((void **)&g_jpeg_na_result[device], 64, hipHostMallocDefault),
                         "hipHostMalloc (result word)")) {
        return false;
    }
    g_jpeg_na_ptr[device] = ptr;
    return true;
}

extern "C" int spice_gpu_jpeg_encode_noalloc(SpiceGpuJpegJob *job)
{
    using namespace spicegpu;
    const int device = current_device();
    if (device < 0 || !g_jpeg_na_ptr[device]) {
        set_error("jpeg (no allocation): call spice_gpu_init first");
        return -1;
    }
    if (!job) {
        set_error("jpeg (no allocation): null job");
        return -1;
    }
    job->out_size = -1;
    uint8_t *const ptr = g_jpeg_na_ptr[device];
    static JpegDevJob dj; /* guarded by the device gate */
    GateHold gate("jpeg (no allocation)");
    if (!gate.ok) {
        return -1;
    }
    if (!check(hipSetDevice(device), "hipSetDevice")) {
        return -1;
    }
    if (!job->pixels || !job->out ||
        !jpeg_fill_job(&dj, job->pixels, job->format, job->width, job->height, job->stride,
                       job->quality, job->dct, job->out, job->out_capacity)) {
        set_error("jpeg (no allocation): unsupported parameters");
        return -1;
    }
    if (dj.nstrips > JPEG_NA_MAX_STRIPS || job->out_capacity > JPEG_NA_MAX_OUT) {
        set_error("jpeg (no allocation): %d strips of 10 MCUs / %u output bytes exceed the maximum of "
                  "%d strips / %u bytes (use spice_gpu_jpeg_encode_batch)",
                  dj.nstrips, job->out_capacity, JPEG_NA_MAX_STRIPS, JPEG_NA_MAX_OUT);
        return -1;
    }
    dj.strip_stride = JPEG_NA_STRIP_STRIDE;
    dj.scan_cap = (job->out_capacity + 64 + 255) & ~255u;
    dj.nchunks_cap = dj.scan_cap / JPEG_CHUNK_BYTES + 1;
    dj.strip_buf = ptr + JPEG_NA_OFF_STRIPS;
    dj.strip_bits = (uint32_t *)(ptr + JPEG_NA_OFF_BITS);
    dj.strip_bitoff = (uint64_t *)(ptr + JPEG_NA_OFF_BITOFF);
    dj.scan_buf = ptr + JPEG_NA_OFF_SCAN;
    dj.chunk_ff = (uint32_t *)(ptr + JPEG_NA_OFF_CHUNKFF);
    dj.result = (int32_t *)(ptr + JPEG_NA_OFF_RESULT);

    const uint64_t t0 = now_ns();
    hipStream_t s = stream();
    const int groups = (dj.strips_per_row + JPEG_GROUP_STRIPS - 1) / JPEG_GROUP_STRIPS * dj.mcu_rows;
    hipLaunchKernelGGL(jpeg_na_put_job, dim3(1), dim3(64), 0, s, dj, (JpegDevJob *)ptr);
    jpeg_launch((const JpegDevJob *)ptr, 1, groups, dj.dct == 0, dj.dct != 0,
                (const uint32_t *)(ptr + JPEG_NA_OFF_HUFF), s, false, nullptr, nullptr);
    int32_t *const landing = g_jpeg_na_result[device];
    if (!check(hipGetLastError(), "jpeg launch") ||
        !check(hipMemcpyAsync(landing, ptr + JPEG_NA_OFF_RESULT, sizeof(int32_t), hipMemcpyDeviceToHost, s),
               "read result") ||
        !check(hipStreamSynchronize(s), "jpeg sync")) {
        return -1;
    }
    const int32_t size = *landing;
    job->out_size = size;
    if (size <= 0) {
        set_error("jpeg (no allocation): output does not fit (%d)", size);
        return -1;
    }
    const uint64_t bpp = job->format == SPICE_GPU_BITMAP_FMT_16BIT ? 2u
                       : (job->format == SPICE_GPU_BITMAP_FMT_24BIT || job->format == JPEG_FMT_RGB24) ? 3u : 4u;
    stat_add(SPICE_GPU_STAT_MJPEG, (uint64_t)dj.width * (uint64_t)dj.height * bpp, (uint64_t)size,
             now_ns() - t0);
    return 0;
}
