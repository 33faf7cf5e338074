/*
 * Device-side job description of the baseline JPEG encoder (jpeg_kernels.hip),
 * filled by jpeg_host.cpp.  One job = one frame of an MJPEG stream
 * (server/mjpeg_encoder.c:771-884) or one still image (server/jpeg_encoder.c:203-248).
 *
 * Pipeline per batch of jobs:
 *   jpeg_strip_kernel   pixels -> YCbCr 4:2:0 in LDS -> DCT -> quantise ->
 *                       Huffman bits; a workgroup takes JPEG_GROUP_STRIPS strips
 *                       of up to JPEG_STRIP_MCUS MCUs, the bits of each land in
 *                       the strip's slot
 *   jpeg_offsets_kernel exclusive scan of the strips' bit counts
 *   jpeg_gather_kernel  concatenates the strips at bit granularity into the
 *                       unstuffed scan, counts 0xFF bytes per 4 KiB chunk
 *   jpeg_finish_kernel  scans the chunk counts, writes header + byte-stuffed
 *                       scan + EOI
 */
#ifndef SPICE_JPEG_DEVICE_H
#define SPICE_JPEG_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#define JPEG_STRIP_MCUS 10            /* MCUs whose bits one wavefront produces */
#define JPEG_GROUP_STRIPS 4           /* strips per workgroup: 40 MCUs + 1 halo MCU = 246 blocks <= 256 lanes */
#define JPEG_STRIP_THREADS 256
#define JPEG_MCU_SLOT_BYTES 1280      /* >= 6 blocks x 1658 bits worst case */
#define JPEG_CHUNK_BYTES 4096         /* unstuffed bytes per 0xFF-count chunk */
#define JPEG_MAX_HEADER 640
#define JPEG_FMT_RGB24 10           /* r,g,b bytes; the SPICE_BITMAP_FMT_* values cover the rest */
#define JPEG_HUFF_WORDS 544          /* AC luma 256, AC chroma 256, DC luma 16, DC chroma 16 */

struct JpegDevJob {
    const uint8_t *pixels;
    int64_t stride;
    int32_t width, height, format, dct;
    int32_t mcu_cols, mcu_rows, strips_per_row, nstrips;
    int32_t yb_w, yb_h;                 /* real luma blocks per row / column */
    int32_t groups_per_row;             /* strip groups (JPEG_GROUP_STRIPS strips) in an MCU row */
    uint32_t groups_rcp;                /* 2^32 / groups_per_row + 1: (group x this) >> 32 is the group's
                                         * MCU row while group x groups_per_row < 2^32 -- a JPEG has at
                                         * most 4096 MCU rows of 103 groups -- (one group per row,
                                         * where this would need 33 bits, is tested for) */
    int32_t div_y[64], div_c[64];       /* jcdctmgr divisors, natural order */
    float rcp_y[64], rcp_c[64];         /* the single-precision number at or just above 1 / divisor */
    float qhalf_y[64], qhalf_c[64];     /* divisor / 2 (integer division) */
    uint8_t *out;
    uint32_t out_cap;
    uint32_t hdr_len;
    uint8_t header[JPEG_MAX_HEADER];
    /* per-job scratch in HBM */
    uint8_t *strip_buf;                 /* nstrips slots of strip_stride bytes */
    uint32_t strip_stride;
    uint32_t *strip_bits;               /* bits produced by each strip */
    uint64_t *strip_bitoff;             /* exclusive scan of the above, + total at [nstrips] */
    uint8_t *scan_buf;                  /* unstuffed scan bytes */
    uint32_t scan_cap;
    uint32_t *chunk_ff;                 /* 0xFF bytes per chunk of the unstuffed scan */
    uint32_t nchunks_cap;
    int32_t *result;                    /* JFIF size or negative error */
};

enum { JPEG_DEV_ERR_NO_SPACE = -3 };

#endif
