/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement of the baseline JPEG encoder the
 * reference obtains from libjpeg (third-party, not vendored by the reference:
 * configure.ac:338-349 accepts whatever libjpeg the system has).
 *
 * Restates the published libjpeg algorithm for the exact configuration the
 * reference selects:
 *   MJPEG  server/mjpeg_encoder.c:771-823  JCS_RGB in, YCbCr 4:2:0 out, standard
 *          tables, JDCT_IFAST, quality from {20..80}, one JFIF per frame
 *   still  server/jpeg_encoder.c:232-246   same, JDCT_ISLOW (libjpeg default)
 * and the reference's own pixel converters (mjpeg_encoder.c:236-259,
 * jpeg_encoder.c:107-158).
 *
 * PARITY: pinned.  tests/golden/jpeg_*.json were produced in the build
 * container by oracle/_ref/libjpegref.so, i.e. the real IJG libjpeg 9d driven
 * with the reference's call sequence (do_fancy_downsampling = FALSE, which is
 * the v6b / libjpeg-turbo downsampler), and this restatement reproduces them
 * byte for byte.  `color` selects the RGB->YCbCr constant set: libjpeg 9 and
 * v6b/libjpeg-turbo round two Cb coefficients differently.
 */
#ifndef JPEG_ORACLE_H
#define JPEG_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { JPGO_DCT_ISLOW = 0, JPGO_DCT_IFAST = 1 };
enum { JPGO_COLOR_V6B = 0, JPGO_COLOR_IJG9 = 1 };
/* same numbering as SPICE_BITMAP_FMT_* (server/red_worker.c:511-512) */
enum { JPGO_FMT_16BIT = 6, JPGO_FMT_24BIT = 7, JPGO_FMT_32BIT = 8, JPGO_FMT_RGBA = 9 };

/* Whole JFIF file.  Returns its size, -1 when `cap` is too small or a
 * parameter is unsupported.  stride may be negative (bottom-up walk). */
long jpgo_encode(const uint8_t *pixels, int width, int height, long stride, int fmt, int quality,
                 int dct, int color, uint8_t *out, size_t cap);

/* Size of everything before the entropy-coded data (SOI .. SOS). */
long jpgo_header(int width, int height, int quality, uint8_t *out, size_t cap);

/* Quantised coefficients, MCU by MCU, 6 blocks (Y00 Y01 Y10 Y11 Cb Cr) of 64
 * int16 in natural (row-major) order; dummy edge blocks included as libjpeg
 * builds them.  Returns the number of MCUs. */
long jpgo_coefficients(const uint8_t *pixels, int width, int height, long stride, int fmt,
                       int quality, int dct, int color, int16_t *coefs, size_t cap_mcus);

#ifdef __cplusplus
}
#endif
#endif
