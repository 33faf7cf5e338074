/*
 * Baseline JPEG: host side.  Builds what libjpeg's jpeg_set_defaults /
 * jpeg_set_quality / jpeg_start_compress would set up for the reference's two
 * call sites -- quantisation tables and DCT divisors, the four standard
 * Huffman tables, the marker segments up to SOS -- and drives the kernels of
 * jpeg_kernels.hip.  Exports the reference's MJPEG and still-JPEG entry points
 * (server/mjpeg_encoder.h, server/jpeg_encoder.h) and the device-resident
 * batch interface.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "jpeg_device.h"
#include "mjpeg_rate.h"
#include "runtime.h"

using namespace spicegpu;

void jpeg_launch(const JpegDevJob *jobs, int njobs, int max_groups, bool any_islow, bool any_ifast,
                 const uint32_t *huff, hipStream_t stream, bool timed, hipEvent_t strips_after,
                 hipEvent_t strips_done);
void jpeg_huff_words(uint32_t *h);

namespace {

/* ITU-T T.81 Annex K tables, as installed by libjpeg's jpeg_set_defaults */
const uint8_t kLumaQ[64] = {
    16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
    14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
    18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
const uint8_t kChromaQ[64] = {
    17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
    24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
const uint8_t kZigzag[64] = {
    0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
    12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
    58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
const uint8_t kDcLumaBits[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const uint8_t kDcChromaBits[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t kDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t kAcLumaBits[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
const uint8_t kAcLumaVals[162] = {
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
const uint8_t kAcChromaBits[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
const uint8_t kAcChromaVals[162] = {
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
/* AA&N scale factors << 14 (libjpeg jcdctmgr.c) */
const int16_t kAanScales[64] = {
    16384, 22725, 21407, 19266, 16384, 12873, 8867,  4520,  22725, 31521, 29692, 26722, 22725,
    17855, 12299, 6270,  21407, 29692, 27969, 25172, 21407, 16819, 11585, 5906,  19266, 26722,
    25172, 22654, 19266, 15137, 10426, 5315,  16384, 22725, 21407, 19266, 16384, 12873, 8867,
    4520,  12873, 17855, 16819, 15137, 12873, 10114, 6967,  3552,  8867,  12299, 11585, 10426,
    8867,  6967,  4799,  2446,  4520,  6270,  5906,  5315,  4520,  3552,  2446,  1247};

/* libjpeg jcparam.c: jpeg_quality_scaling + jpeg_add_quant_table(force_baseline) */
void quant_table(const uint8_t *base, int quality, uint16_t *q)
{
    quality = quality <= 0 ? 1 : (quality > 100 ? 100 : quality);
    int scale = quality < 50 ? 5000 / quality : 200 - quality * 2;
    for (int i = 0; i < 64; i++) {
        long t = ((long)base[i] * scale + 50L) / 100L;
        t = t <= 0 ? 1 : (t > 255 ? 255 : t);
        q[i] = (uint16_t)t;
    }
}

/* canonical codes of a (bits, vals) table, packed size << 16 | code */
void pack_huff(const uint8_t *bits, const uint8_t *vals, uint32_t *out, int nout)
{
    memset(out, 0, sizeof(uint32_t) * (size_t)nout);
    unsigned code = 0;
    int k = 0;
    for (int l = 1; l <= 16; l++) {
        for (int i = 0; i < bits[l - 1]; i++) {
            out[vals[k]] = ((uint32_t)l << 16) | code;
            code++;
            k++;
        }
        code <<= 1;
    }
}

struct ByteSink {
    uint8_t *p;
    size_t n;
    void b(unsigned v) { p[n++] = (uint8_t)v; }
    void w(unsigned v)
    {
        b(v >> 8);
        b(v);
    }
};

void put_dht(ByteSink &s, int idx, const uint8_t *bits, const uint8_t *vals)
{
    int n = 0;
    for (int i = 0; i < 16; i++) {
        n += bits[i];
    }
    s.w(0xffc4);
    s.w((unsigned)(19 + n));
    s.b((unsigned)idx);
    for (int i = 0; i < 16; i++) s.b(bits[i]);
    for (int i = 0; i < n; i++) s.b(vals[i]);
}

/* libjpeg jcmarker.c: SOI, JFIF APP0, two DQT, SOF0, four DHT, SOS */
size_t build_header(int width, int height, const uint16_t *qy, const uint16_t *qc, uint8_t *out)
{
    static const uint8_t app0[] = {0xff, 0xe0, 0, 16, 'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0};
    ByteSink s{out, 0};
    s.w(0xffd8);
    for (uint8_t v : app0) s.b(v);
    for (int t = 0; t < 2; t++) {
        const uint16_t *q = t ? qc : qy;
        s.w(0xffdb);
        s.w(67);
        s.b((unsigned)t);
        for (int i = 0; i < 64; i++) s.b(q[kZigzag[i]]);
    }
    s.w(0xffc0);
    s.w(17);
    s.b(8);
    s.w((unsigned)height);
    s.w((unsigned)width);
    s.b(3);
    s.b(1); s.b(0x22); s.b(0);
    s.b(2); s.b(0x11); s.b(1);
    s.b(3); s.b(0x11); s.b(1);
    put_dht(s, 0x00, kDcLumaBits, kDcVals);
    put_dht(s, 0x10, kAcLumaBits, kAcLumaVals);
    put_dht(s, 0x01, kDcChromaBits, kDcVals);
    put_dht(s, 0x11, kAcChromaBits, kAcChromaVals);
    s.w(0xffda);
    s.w(12);
    s.b(3);
    s.b(1); s.b(0x00);
    s.b(2); s.b(0x11);
    s.b(3); s.b(0x11);
    s.b(0); s.b(63); s.b(0);
    return s.n;
}

int format_bpp(int fmt)
{
    switch (fmt) {
    case SPICE_GPU_BITMAP_FMT_16BIT: return 2;
    case SPICE_GPU_BITMAP_FMT_24BIT: return 3;
    case JPEG_FMT_RGB24: return 3;
    case SPICE_GPU_BITMAP_FMT_32BIT:
    case SPICE_GPU_BITMAP_FMT_RGBA: return 4;
    default: return 0;
    }
}

/* the single-precision number at or just above 1 / d */
static float recip_not_below(int32_t d)
{
    float r = (float)(1.0 / (double)d);
    if ((double)r * (double)d < 1.0) {
        r = nextafterf(r, 2.0f);
    }
    return r;
}

/* What a job's record holds besides its buffers depends on six numbers only, and a
 * stream codes frame after frame with the same six: the quantisation tables, their
 * reciprocals and the 600-byte header cost a few microseconds to make, a batch of
 * sixteen frames more than a fifth of what the GPU then needs for it.  Every thread
 * keeps the last few records it made. */
struct JobTemplate {
    bool valid;
    int format, width, height, quality, dct;
    long stride;
    JpegDevJob job; /* pixels, out, out_cap and the per-batch scratch pointers unset */
};
static thread_local JobTemplate t_templates[4];
static thread_local unsigned t_template_next;

static bool build_job(JpegDevJob *j, int format, int width, int height, long stride, int quality, int dct);

/* Fills everything of a device job that depends only on the parameters. */
bool fill_job(JpegDevJob *j, const void *pixels, int format, int width, int height, long stride,
              int quality, int dct, void *out, uint32_t out_cap)
{
    const JobTemplate *hit = nullptr;
    for (const JobTemplate &t : t_templates) {
        if (t.valid && t.format == format && t.width == width && t.height == height && t.stride == stride &&
            t.quality == quality && t.dct == (dct ? 1 : 0)) {
            hit = &t;
            break;
        }
    }
    if (!hit) {
        JobTemplate &t = t_templates[t_template_next++ % 4];
        t.valid = false;
        if (!build_job(&t.job, format, width, height, stride, quality, dct)) {
            return false;
        }
        t.format = format;
        t.width = width;
        t.height = height;
        t.stride = stride;
        t.quality = quality;
        t.dct = dct ? 1 : 0;
        t.valid = true;
        hit = &t;
    }
    *j = hit->job;
    j->pixels = (const uint8_t *)pixels;
    j->out = (uint8_t *)out;
    j->out_cap = out_cap;
    return true;
}

static bool build_job(JpegDevJob *j, int format, int width, int height, long stride, int quality, int dct)
{
    if (!format_bpp(format) || width <= 0 || height <= 0 || width > 65535 || height > 65535) {
        return false;
    }
    memset(j, 0, sizeof(*j));
    j->stride = stride;
    j->width = width;
    j->height = height;
    j->format = format;
    j->dct = dct ? 1 : 0;
    j->mcu_cols = (width + 15) / 16;
    j->mcu_rows = (height + 15) / 16;
    j->strips_per_row = (j->mcu_cols + JPEG_STRIP_MCUS - 1) / JPEG_STRIP_MCUS;
    j->nstrips = j->strips_per_row * j->mcu_rows;
    j->yb_w = (width + 7) / 8;
    j->yb_h = (height + 7) / 8;
    j->groups_per_row = (j->strips_per_row + JPEG_GROUP_STRIPS - 1) / JPEG_GROUP_STRIPS;
    j->groups_rcp = (uint32_t)((1ull << 32) / (uint64_t)j->groups_per_row + 1);
    uint16_t qy[64], qc[64];
    quant_table(kLumaQ, quality, qy);
    quant_table(kChromaQ, quality, qc);
    for (int i = 0; i < 64; i++) {
        /* libjpeg jcdctmgr.c start_pass_fdctmgr */
        if (dct) {
            j->div_y[i] = (int32_t)(((int32_t)qy[i] * kAanScales[i] + (1 << 10)) >> 11);
            j->div_c[i] = (int32_t)(((int32_t)qc[i] * kAanScales[i] + (1 << 10)) >> 11);
        } else {
            j->div_y[i] = (int32_t)qy[i] << 3;
            j->div_c[i] = (int32_t)qc[i] << 3;
        }
        /* the kernel quantises in single precision and truncates the product: its
         * reciprocal must not lie below 1/d */
        j->rcp_y[i] = recip_not_below(j->div_y[i]);
        j->rcp_c[i] = recip_not_below(j->div_c[i]);
        j->qhalf_y[i] = (float)(j->div_y[i] >> 1);
        j->qhalf_c[i] = (float)(j->div_c[i] >> 1);
    }
    j->hdr_len = (uint32_t)build_header(width, height, qy, qc, j->header);
    return true;
}

} // namespace

/* for the entry point that allocates nothing (jpeg_kernels.hip) */
bool jpeg_fill_job(JpegDevJob *j, const void *pixels, int format, int width, int height, long stride,
                   int quality, int dct, void *out, uint32_t out_cap)
{
    return fill_job(j, pixels, format, width, height, stride, quality, dct, out, out_cap);
}

/* the four standard Huffman tables as the kernels read them: AC luma, AC chroma,
 * DC luma, DC chroma */
void jpeg_huff_words(uint32_t *h)
{
    pack_huff(kAcLumaBits, kAcLumaVals, h, 256);
    pack_huff(kAcChromaBits, kAcChromaVals, h + 256, 256);
    pack_huff(kDcLumaBits, kDcVals, h + 512, 16);
    pack_huff(kDcChromaBits, kDcVals, h + 528, 16);
}

namespace {

/* Library-owned device scratch and stream of one batch. */
struct JpegFlight {
    Staging staging;      /* JpegDevJob array + results */
    DevBuffer scratch;
    uint32_t *d_huff = nullptr;
    size_t results_off = 0;
    int njobs = 0;
    bool active = false;  /* claimed by a submit, until its finish; g_jf_mutex guards it */
    std::thread::id owner;
    uint64_t seq = 0;     /* order of submission */
    uint64_t t0 = 0;
    uint64_t raw_bytes = 0;
    int stat_codec = SPICE_GPU_STAT_MJPEG;
    hipEvent_t strips_done = nullptr; /* recorded behind this flight's strip kernels */
    OwnStream own;
};
/* Two batches may be in flight, the second submitted while the first still runs: each has
 * its own stream, job upload area and scratch, so the kernels of one overlap the host's
 * work for, and the short bit-stream kernels of, the other.  A slot belongs to the thread
 * whose submit claimed it until that thread's finish; claiming, enqueuing and releasing
 * happen under g_jf_mutex, which no call keeps when it returns -- a thread that pipelines
 * batches holds up nobody but a third batch.  The synchronous entry points have a flight
 * of their own (g_jf_sync) on the library's stream(), used under the device gate. */
const int kJpegFlights = 2;
JpegFlight g_jfs[kJpegFlights];
JpegFlight g_jf_sync;
std::mutex g_jf_mutex;
std::condition_variable g_jf_cv;
uint64_t g_jf_seq = 0;

hipStream_t flight_stream(const JpegFlight &f)
{
    return &f == &g_jf_sync ? stream() : f.own.st;
}

bool ensure_huff(JpegFlight &f)
{
    if (f.d_huff) {
        return true;
    }
    uint32_t h[JPEG_HUFF_WORDS];
    jpeg_huff_words(h);
    if (!check(hipMalloc((void **)&f.d_huff, sizeof(h)), "hipMalloc huff")) {
        return false;
    }
    return check(hipMemcpy(f.d_huff, h, sizeof(h), hipMemcpyHostToDevice), "upload huff");
}

size_t align256(size_t v)
{
    return (v + 255) & ~(size_t)255;
}

/* Enqueues a batch into a flight the caller owns (g_jf_mutex held for g_jfs[], the device
 * gate for g_jf_sync); jobs[] filled with fill_job. */
bool submit_locked(JpegFlight &f, std::vector<JpegDevJob> &jobs)
{
    int n = (int)jobs.size();
    if (!ensure_huff(f)) {
        return false;
    }
    size_t need = 0;
    int max_groups = 0;
    bool any_islow = false, any_ifast = false;
    for (JpegDevJob &j : jobs) {
        j.strip_stride = JPEG_MCU_SLOT_BYTES * JPEG_STRIP_MCUS;
        j.scan_cap = (uint32_t)align256(j.out_cap + 64);
        j.nchunks_cap = j.scan_cap / JPEG_CHUNK_BYTES + 1;
        need += align256((size_t)j.nstrips * j.strip_stride + 64);
        need += align256(sizeof(uint32_t) * (size_t)j.nstrips);
        need += align256(sizeof(uint64_t) * ((size_t)j.nstrips + 1));
        need += align256(j.scan_cap + 64);
        need += align256(sizeof(uint32_t) * j.nchunks_cap);
        const int groups = (j.strips_per_row + JPEG_GROUP_STRIPS - 1) / JPEG_GROUP_STRIPS * j.mcu_rows;
        max_groups = groups > max_groups ? groups : max_groups;
        (j.dct ? any_ifast : any_islow) = true;
    }
    size_t jobs_bytes = align256(sizeof(JpegDevJob) * (size_t)n);
    if (!f.scratch.reserve(need) || !f.staging.reserve(jobs_bytes + sizeof(int32_t) * (size_t)n)) {
        return false;
    }
    uint8_t *p = f.scratch.ptr;
    JpegDevJob *hj = (JpegDevJob *)f.staging.host;
    for (int k = 0; k < n; k++) {
        JpegDevJob &j = jobs[(size_t)k];
        j.strip_buf = p;
        p += align256((size_t)j.nstrips * j.strip_stride + 64);
        j.strip_bits = (uint32_t *)p;
        p += align256(sizeof(uint32_t) * (size_t)j.nstrips);
        j.strip_bitoff = (uint64_t *)p;
        p += align256(sizeof(uint64_t) * ((size_t)j.nstrips + 1));
        j.scan_buf = p;
        p += align256(j.scan_cap + 64);
        j.chunk_ff = (uint32_t *)p;
        p += align256(sizeof(uint32_t) * j.nchunks_cap);
        j.result = (int32_t *)(f.staging.dev + jobs_bytes) + k;
        hj[k] = j;
    }
    hipStream_t s = flight_stream(f);
    if (!check(hipMemcpyAsync(f.staging.dev, f.staging.host, jobs_bytes, hipMemcpyHostToDevice, s),
               "upload jpeg jobs")) {
        return false;
    }
    if (!f.strips_done &&
        !check(hipEventCreate(&f.strips_done), "hipEventCreate")) {
        return false;
    }
    /* the other slot's strip kernels first, if it has a batch in flight (two strip kernels
     * side by side share the CUs, each lasts longer and a per-launch duration says nothing) */
    hipEvent_t after = nullptr;
    if (&f != &g_jf_sync) {
        JpegFlight &other = g_jfs[&f == &g_jfs[0] ? 1 : 0];
        after = other.active ? other.strips_done : nullptr;
    }
    jpeg_launch((const JpegDevJob *)f.staging.dev, n, max_groups, any_islow, any_ifast, f.d_huff, s, true,
                after, f.strips_done);
    if (!check(hipGetLastError(), "jpeg launch")) {
        return false;
    }
    f.results_off = jobs_bytes;
    f.njobs = n;
    return check(hipMemcpyAsync(f.staging.host + jobs_bytes, f.staging.dev + jobs_bytes,
                                sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, s),
                 "download jpeg results");
}

/* Synchronous single image from a host frame (top-down, contiguous rows). */
struct HostPath {
    PinnedBuffer frame;   /* rows gathered on the host */
    DevBuffer d_frame;
    DevBuffer d_out;
    PinnedBuffer out;
};

/* Returns the JFIF size (> 0) with the bytes in hp.out.ptr, or 0. */
size_t encode_host_frame(HostPath &hp, int format, int width, int height, int quality, int dct,
                         int stat_codec)
{
    uint64_t t0 = now_ns();
    int bpp = format_bpp(format);
    size_t row = (size_t)width * (size_t)bpp;
    size_t bytes = row * (size_t)height;
    size_t cap = (size_t)width * (size_t)height * 3 + 4096; /* far above any real frame */
    cap = cap > 0x7fff0000u ? 0x7fff0000u : cap;
    GateHold dev("jpeg");
    if (!dev.ok) {
        return 0;
    }
    if (!hp.d_frame.reserve(bytes + 64) || !hp.d_out.reserve(cap) || !hp.out.reserve(cap)) {
        return 0;
    }
    std::vector<JpegDevJob> jobs(1);
    if (!fill_job(&jobs[0], hp.d_frame.ptr, format, width, height, (long)row, quality, dct,
                  hp.d_out.ptr, (uint32_t)cap)) {
        set_error("jpeg: unsupported parameters");
        return 0;
    }
    hipStream_t s = stream();
    if (!check(hipMemcpyAsync(hp.d_frame.ptr, hp.frame.ptr, bytes, hipMemcpyHostToDevice, s),
               "upload frame") ||
        !submit_locked(g_jf_sync, jobs) || !check(hipStreamSynchronize(s), "jpeg sync")) {
        return 0;
    }
    int32_t size = *(int32_t *)(g_jf_sync.staging.host + g_jf_sync.results_off);
    if (size <= 0) {
        set_error("jpeg: output does not fit (%d)", size);
        return 0;
    }
    if (!check(hipMemcpyAsync(hp.out.ptr, hp.d_out.ptr, (size_t)size, hipMemcpyDeviceToHost, s),
               "download jfif") ||
        !check(hipStreamSynchronize(s), "jpeg sync")) {
        return 0;
    }
    stat_add(stat_codec, bytes, (uint64_t)size, now_ns() - t0);
    return (size_t)size;
}

} // namespace

struct MJpegEncoder {
    HostPath hp;
    int format = 0, width = 0, height = 0, bpp = 0;
    int rows_in = 0;
    bool in_frame = false;
    bool rate_only = false; /* no codec behind it, see spice_gpu_mjpeg_rate_only_new */
    uint8_t **dest = nullptr;
    size_t *dest_len = nullptr;
    spicegpu::MjpegRate rate;
    int frame_quality = 0;
    uint64_t starting_bit_rate = 0;
    uint64_t sum_quality = 0;
    uint32_t num_frames = 0;
};

struct StillJpeg {
    JpegEncoderUsrContext *usr;
    HostPath hp;
};

extern "C" {

/* ------------------------------------------------------------------- MJPEG */
MJpegEncoder *mjpeg_encoder_new(int bit_rate_control, uint64_t starting_bit_rate,
                                MJpegEncoderRateControlCbs *cbs, void *opaque)
{
    /* server/mjpeg_encoder.c:192 asserts this */
    if (bit_rate_control && !(cbs && cbs->get_roundtrip_ms && cbs->get_source_fps)) {
        set_error("mjpeg: rate control needs get_roundtrip_ms and get_source_fps");
        return nullptr;
    }
    if (!ensure_init()) {
        return nullptr;
    }
    MJpegEncoder *e = new MJpegEncoder();
    e->starting_bit_rate = starting_bit_rate;
    e->rate.init(bit_rate_control != 0, starting_bit_rate, cbs, opaque);
    return e;
}

MJpegEncoder *spice_gpu_mjpeg_rate_only_new(uint64_t starting_bit_rate,
                                            MJpegEncoderRateControlCbs *cbs, void *opaque,
                                            uint64_t (*clock_fn)(void *), void *clock_opaque)
{
    if (!(cbs && cbs->get_roundtrip_ms && cbs->get_source_fps)) {
        set_error("mjpeg: rate control needs get_roundtrip_ms and get_source_fps");
        return nullptr;
    }
    MJpegEncoder *e = new MJpegEncoder();
    e->rate_only = true;
    e->starting_bit_rate = starting_bit_rate;
    e->rate.set_clock(clock_fn, clock_opaque);
    e->rate.init(true, starting_bit_rate, cbs, opaque);
    return e;
}

int spice_gpu_mjpeg_rate_feed(MJpegEncoder *e, uint32_t frame_mm_time,
                              const uint64_t *size_by_quality_id, int *quality)
{
    if (!e->rate.admit_frame(frame_mm_time)) {
        return MJPEG_ENCODER_FRAME_DROP;
    }
    e->frame_quality = e->rate.quality();
    e->num_frames++;
    e->sum_quality += (uint64_t)e->frame_quality;
    if (quality) {
        *quality = e->frame_quality;
    }
    const uint64_t enc_size = size_by_quality_id[(e->frame_quality - 20) / 10];
    if (enc_size) {
        e->rate.frame_coded(enc_size);
    } else {
        e->rate.frame_aborted();
    }
    return MJPEG_ENCODER_FRAME_ENCODE_START;
}

void spice_gpu_mjpeg_set_clock(MJpegEncoder *e, uint64_t (*clock_fn)(void *), void *opaque)
{
    e->rate.set_clock(clock_fn, opaque);
}

void spice_gpu_mjpeg_rate_state(MJpegEncoder *e, SpiceGpuMjpegRateState *out)
{
    e->rate.snapshot(out);
}

void mjpeg_encoder_client_stream_report(MJpegEncoder *e, uint32_t num_frames, uint32_t num_drops,
                                        uint32_t start_frame_mm_time, uint32_t end_frame_mm_time,
                                        int32_t end_frame_delay, uint32_t audio_delay)
{
    e->rate.client_report(num_frames, num_drops, start_frame_mm_time, end_frame_mm_time,
                          end_frame_delay, audio_delay);
}

void mjpeg_encoder_notify_server_frame_drop(MJpegEncoder *e)
{
    e->rate.server_dropped_frame();
}

void mjpeg_encoder_destroy(MJpegEncoder *e)
{
    if (!e) {
        return;
    }
    e->hp.frame.release();
    e->hp.d_frame.release();
    e->hp.d_out.release();
    e->hp.out.release();
    delete e;
}

uint8_t mjpeg_encoder_get_bytes_per_pixel(MJpegEncoder *e)
{
    return (uint8_t)e->bpp;
}

int mjpeg_encoder_start_frame(MJpegEncoder *e, int format, int width, int height, uint8_t **dest,
                              size_t *dest_len, uint32_t frame_mm_time)
{
    if (e->rate_only) {
        set_error("mjpeg: this encoder is a rate controller only");
        return MJPEG_ENCODER_FRAME_UNSUPPORTED;
    }
    /* the rate decision comes first, as in server/mjpeg_encoder.c:736-769 */
    if (!e->rate.admit_frame(frame_mm_time)) {
        return MJPEG_ENCODER_FRAME_DROP;
    }
    /* server/mjpeg_encoder.c:775-801 */
    if (format != SPICE_GPU_BITMAP_FMT_32BIT && format != SPICE_GPU_BITMAP_FMT_RGBA &&
        format != SPICE_GPU_BITMAP_FMT_24BIT && format != SPICE_GPU_BITMAP_FMT_16BIT) {
        return MJPEG_ENCODER_FRAME_UNSUPPORTED;
    }
    if (width <= 0 || height <= 0 || width > 65535 || height > 65535 || !dest || !dest_len ||
        !ensure_init()) {
        return MJPEG_ENCODER_FRAME_UNSUPPORTED;
    }
    e->format = format;
    e->bpp = format_bpp(format);
    e->width = width;
    e->height = height;
    e->rows_in = 0;
    e->dest = dest;
    e->dest_len = dest_len;
    if (!e->hp.frame.reserve((size_t)width * (size_t)e->bpp * (size_t)height)) {
        return MJPEG_ENCODER_FRAME_UNSUPPORTED;
    }
    e->in_frame = true;
    e->frame_quality = e->rate.quality(); /* :821 */
    e->num_frames++;
    e->sum_quality += (uint64_t)e->frame_quality;
    return MJPEG_ENCODER_FRAME_ENCODE_START;
}

int mjpeg_encoder_encode_scanline(MJpegEncoder *e, uint8_t *src_pixels, size_t image_width)
{
    if (!e->in_frame || e->rows_in >= e->height || image_width != (size_t)e->width) {
        e->rate.frame_aborted();
        return 0;
    }
    size_t row = (size_t)e->width * (size_t)e->bpp;
    memcpy(e->hp.frame.ptr + row * (size_t)e->rows_in, src_pixels, row);
    e->rows_in++;
    return 1;
}

size_t mjpeg_encoder_end_frame(MJpegEncoder *e)
{
    if (!e->in_frame || e->rows_in != e->height) {
        set_error("mjpeg: end_frame after %d of %d scanlines", e->rows_in, e->height);
        e->in_frame = false;
        return 0;
    }
    e->in_frame = false;
    /* server/mjpeg_encoder.c:819-823: defaults, IFAST, quality, one JFIF per frame */
    size_t n = encode_host_frame(e->hp, e->format, e->width, e->height, e->frame_quality,
                                 SPICE_GPU_DCT_IFAST, SPICE_GPU_STAT_MJPEG);
    if (!n) {
        e->rate.frame_aborted();
        return 0;
    }
    /* spice_jpeg_mem_dest semantics (server/mjpeg_encoder.c:281-359): the buffer is
     * malloc()ed, grown by doubling, and handed back through *dest / *dest_len */
    size_t cap = (*e->dest && *e->dest_len) ? *e->dest_len : 4096;
    if (!*e->dest || !*e->dest_len || cap < n) {
        while (cap < n) {
            cap *= 2;
        }
        uint8_t *nb = (uint8_t *)malloc(cap);
        if (!nb) {
            set_error("mjpeg: out of memory");
            e->rate.frame_aborted();
            return 0;
        }
        free(*e->dest);
        *e->dest = nb;
        *e->dest_len = cap;
    }
    memcpy(*e->dest, e->hp.out.ptr, n);
    e->rate.frame_coded(n);
    return n;
}

uint64_t mjpeg_encoder_get_bit_rate(MJpegEncoder *e)
{
    return e->rate.bit_rate(); /* byte_rate * 8, server/mjpeg_encoder.c:1260-1263 */
}

void mjpeg_encoder_get_stats(MJpegEncoder *e, MJpegEncoderStats *stats)
{
    stats->starting_bit_rate = e->starting_bit_rate;
    stats->cur_bit_rate = mjpeg_encoder_get_bit_rate(e);
    stats->avg_quality = e->num_frames ? (double)e->sum_quality / e->num_frames : 0.0;
}

/* -------------------------------------------------------------- still JPEG */
JpegEncoderContext *jpeg_encoder_create(JpegEncoderUsrContext *usr)
{
    if (!usr || !usr->more_space || !usr->more_lines || !ensure_init()) {
        return nullptr; /* server/jpeg_encoder.c:81-83 */
    }
    StillJpeg *s = new StillJpeg();
    s->usr = usr;
    return (JpegEncoderContext *)s;
}

void jpeg_encoder_destroy(JpegEncoderContext *encoder)
{
    StillJpeg *s = (StillJpeg *)encoder;
    if (!s) {
        return;
    }
    s->hp.frame.release();
    s->hp.d_frame.release();
    s->hp.d_out.release();
    s->hp.out.release();
    delete s;
}

int jpeg_encode(JpegEncoderContext *jpeg, int quality, JpegEncoderImageType type, int width,
                int height, uint8_t *lines, unsigned int num_lines, int stride, uint8_t *io_ptr,
                unsigned int num_io_bytes)
{
    StillJpeg *s = (StillJpeg *)jpeg;
    int format;
    switch (type) {
    case JPEG_IMAGE_TYPE_RGB16: format = SPICE_GPU_BITMAP_FMT_16BIT; break;
    case JPEG_IMAGE_TYPE_RGB24: format = JPEG_FMT_RGB24; break;
    case JPEG_IMAGE_TYPE_BGR24: format = SPICE_GPU_BITMAP_FMT_24BIT; break;
    case JPEG_IMAGE_TYPE_BGRX32: format = SPICE_GPU_BITMAP_FMT_32BIT; break;
    default:
        set_error("jpeg: bad image type");
        return 0;
    }
    if (width <= 0 || height <= 0 || !ensure_init()) {
        return 0;
    }
    int bpp = format_bpp(format);
    size_t row = (size_t)width * (size_t)bpp;
    if (!s->hp.frame.reserve(row * (size_t)height)) {
        return 0;
    }
    /* do_jpeg_encode, server/jpeg_encoder.c:176-201: walk `stride` (possibly
     * negative) from `lines`, refill through more_lines when a chunk ends */
    {
        uint8_t *cur = lines;
        uint8_t *end = lines + (long)stride * (long)num_lines;
        for (int y = 0; y < height; y++, cur += stride) {
            if (cur == end) {
                int n = s->usr->more_lines(s->usr, &cur);
                if (n <= 0) {
                    set_error("jpeg: more lines failed");
                    return 0;
                }
                end = cur + (long)n * (long)stride;
            }
            memcpy(s->hp.frame.ptr + row * (size_t)y, cur, row);
        }
    }
    size_t n = encode_host_frame(s->hp, format, width, height, quality, SPICE_GPU_DCT_ISLOW,
                                 SPICE_GPU_STAT_JPEG);
    if (!n) {
        return 0;
    }
    /* dest_mgr callbacks, server/jpeg_encoder.c:44-76 */
    size_t left = n, room = num_io_bytes;
    const uint8_t *src = s->hp.out.ptr;
    uint8_t *dst = io_ptr;
    while (left) {
        if (room == 0) {
            int got = s->usr->more_space(s->usr, &dst);
            if (got <= 0) {
                set_error("jpeg: not enough space");
                return 0;
            }
            room = (size_t)got;
        }
        size_t k = left < room ? left : room;
        memcpy(dst, src, k);
        src += k;
        dst += k;
        left -= k;
        room -= k;
    }
    return (int)n;
}

/* ----------------------------------------------------------- batch interface */
size_t spice_gpu_jpeg_out_bound(int width, int height)
{
    /* 6 blocks x 1658 bits per MCU, every byte stuffed, plus markers */
    size_t mcus = (size_t)((width + 15) / 16) * (size_t)((height + 15) / 16);
    return JPEG_MAX_HEADER + mcus * 1244 * 2 + 16;
}

int spice_gpu_jpeg_submit(SpiceGpuJpegJob *jobs, int njobs)
{
    if (!ensure_init()) {
        return -1;
    }
    if (njobs <= 0) {
        set_error("jpeg batch: no jobs");
        return -1;
    }
    std::vector<JpegDevJob> dj((size_t)njobs);
    uint64_t raw_bytes = 0;
    for (int k = 0; k < njobs; k++) {
        SpiceGpuJpegJob &j = jobs[k];
        j.out_size = -1;
        if (!fill_job(&dj[(size_t)k], j.pixels, j.format, j.width, j.height, j.stride, j.quality,
                      j.dct, j.out, j.out_capacity)) {
            set_error("jpeg batch job %d: unsupported parameters", k);
            return -1;
        }
        raw_bytes += (uint64_t)j.width * (uint64_t)j.height * (uint64_t)format_bpp(j.format);
    }
    /* a free flight slot: a thread may hold both, a third submit is refused, and while
     * other threads hold them this one waits for a finish */
    std::unique_lock<std::mutex> lk(g_jf_mutex);
    const std::thread::id me = std::this_thread::get_id();
    JpegFlight *fp = nullptr;
    for (;;) {
        int mine = 0;
        fp = nullptr;
        for (JpegFlight &c : g_jfs) {
            if (!c.active) {
                fp = fp ? fp : &c;
            } else if (c.owner == me) {
                mine++;
            }
        }
        if (mine >= kJpegFlights) {
            set_error("jpeg batch: two batches are in flight already (finish one first)");
            return -1;
        }
        if (fp) {
            break;
        }
        g_jf_cv.wait(lk);
    }
    JpegFlight &f = *fp;
    f.raw_bytes = raw_bytes;
    f.t0 = now_ns();
    if (!f.own.ensure() || !submit_locked(f, dj)) {
        if (f.own.st) {
            hipStreamSynchronize(f.own.st);
        }
        return -1;
    }
    f.active = true;
    f.owner = me;
    f.seq = ++g_jf_seq;
    return 0;
}

/* Completes the calling thread's OLDEST batch in flight: jobs / njobs are that batch's. */
int spice_gpu_jpeg_finish(SpiceGpuJpegJob *jobs, int njobs)
{
    JpegFlight *fp = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_jf_mutex);
        const std::thread::id me = std::this_thread::get_id();
        for (JpegFlight &c : g_jfs) {
            if (c.active && c.owner == me && (!fp || c.seq < fp->seq)) {
                fp = &c;
            }
        }
        if (!fp) {
            set_error("jpeg batch: finish without a matching submit");
            return -1;
        }
        if (njobs != fp->njobs) {
            set_error("jpeg batch: finish does not match the oldest batch in flight (%d jobs, not %d)",
                      fp->njobs, njobs);
            return -1;
        }
    }
    JpegFlight &f = *fp;
    bool ok = check(hipStreamSynchronize(flight_stream(f)), "jpeg batch sync");
    const int32_t *res = (const int32_t *)(f.staging.host + f.results_off);
    int rc = ok ? 0 : -1;
    for (int k = 0; k < njobs; k++) {
        jobs[k].out_size = ok ? res[k] : -1;
        if (jobs[k].out_size <= 0) {
            if (ok) {
                set_error("jpeg batch job %d: output does not fit (%d)", k, jobs[k].out_size);
            }
            rc = -1;
        }
    }
    if (rc == 0) {
        uint64_t dt = now_ns() - f.t0;
        for (int k = 0; k < njobs; k++) {
            stat_add(SPICE_GPU_STAT_MJPEG, f.raw_bytes / (uint64_t)njobs, (uint64_t)jobs[k].out_size,
                     dt / (uint64_t)njobs);
        }
    }
    {
        std::lock_guard<std::mutex> lk(g_jf_mutex);
        f.active = false;
        g_jf_cv.notify_all();
    }
    return rc;
}

int spice_gpu_jpeg_encode_batch(SpiceGpuJpegJob *jobs, int njobs)
{
    {
        std::lock_guard<std::mutex> lk(g_jf_mutex);
        const std::thread::id me = std::this_thread::get_id();
        for (JpegFlight &c : g_jfs) {
            if (c.active && c.owner == me) {
                /* finish would complete the older batch, not this one */
                set_error("jpeg batch: this thread has a batch in flight (finish it first)");
                return -1;
            }
        }
    }
    if (spice_gpu_jpeg_submit(jobs, njobs) != 0) {
        return -1;
    }
    return spice_gpu_jpeg_finish(jobs, njobs);
}

} // extern "C"
