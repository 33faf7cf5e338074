/*
 * Host side of the codec choice: the graduality score of a bitmap (sample
 * geometry, launch, the final division and thresholds) and
 * red_compress_image's decision (server/red_worker.c:5904-5967, 6411-6525).
 */
#include <hip/hip_runtime.h>
#include <string.h>

#include <map>
#include <utility>
#include <vector>

#include "gradual_device.h"
#include "runtime.h"

using namespace spicegpu;

namespace {

const int kSampleJump = 15;            /* SAMPLE_JUMP, red_bitmap_utils.h:54 */
const double kHighRgb24 = -0.03;       /* GRADUAL_HIGH_RGB24_TH, red_worker.c:5904 */
const double kHighRgb16 = 0;           /* GRADUAL_HIGH_RGB16_TH */
const double kMedium = 0.002;          /* GRADUAL_MEDIUM_SCORE_TH */

int format_bytes(int format)
{
    switch (format) {
    case SPICE_GPU_BITMAP_FMT_16BIT:
        return 2;
    case SPICE_GPU_BITMAP_FMT_24BIT:
        return 3;
    case SPICE_GPU_BITMAP_FMT_32BIT:
    case SPICE_GPU_BITMAP_FMT_RGBA:
        return 4;
    default:
        return 0;
    }
}

/* compute_lines_gradual_score's walk (red_bitmap_utils.h:119-147) reduced to
 * the column of the first sample of every row but the last.  Returns the
 * number of samples. */
uint64_t sample_rows(int width, uint32_t rows, int jump, uint32_t *first_col)
{
    const uint64_t w = (uint64_t)width;
    const uint64_t last_line = (uint64_t)(rows - 1) * w;
    uint64_t p = w / 2, n = 0;
    for (uint32_t r = 0; r + 1 < rows; r++) {
        const uint64_t row_begin = (uint64_t)r * w, row_end = row_begin + w;
        if (p >= row_end || p >= last_line) {
            first_col[r] = GRADUAL_NO_SAMPLE;
            continue;
        }
        const uint64_t e = p - row_begin;
        const uint64_t k = (w - 1 - e) / (uint64_t)jump + 1; /* samples in this row */
        uint64_t last = e + (k - 1) * (uint64_t)jump;
        if (last + 1 == w) {
            last--; /* the last pixel of a row has no right neighbour, :137-139 */
        }
        first_col[r] = (uint32_t)e;
        n += k;
        p = row_begin + last + (uint64_t)jump;
    }
    return n;
}

BitmapGradualType level_of(int format, double score)
{
    if (format == SPICE_GPU_BITMAP_FMT_16BIT) {
        if (score < kHighRgb16) {
            return BITMAP_GRADUAL_HIGH;
        }
    } else if (score < kHighRgb24) {
        return BITMAP_GRADUAL_HIGH;
    }
    return score < kMedium ? BITMAP_GRADUAL_MEDIUM : BITMAP_GRADUAL_LOW;
}

Staging g_stage;      /* jobs | chunks | first-sample columns, uploaded in one copy */
Staging g_results;    /* 2 x int64 per job, zeroed on upload, read back */

} // namespace

extern "C" {

int spice_gpu_graduality_batch(SpiceGpuGradualJob *jobs, int njobs)
{
    if (!ensure_init()) {
        return -1;
    }
    if (njobs <= 0) {
        set_error("graduality batch: no jobs");
        return -1;
    }
    std::vector<GradualDevJob> dj((size_t)njobs);
    std::vector<GradualDevChunk> dc;
    std::vector<uint32_t> pool;
    std::map<std::pair<int, uint32_t>, uint32_t> tables; /* (width, rows) -> offset into pool */
    std::vector<int> trivial((size_t)njobs, 0);          /* chunks scored 1.0 with one sample, :127-131 */

    for (int k = 0; k < njobs; k++) {
        SpiceGpuGradualJob &j = jobs[k];
        j.level = BITMAP_GRADUAL_INVALID;
        if (!format_bytes(j.format) || j.width <= 0 || j.num_chunks <= 0 ||
            j.num_chunks > SPICE_GPU_GRADUAL_MAX_CHUNKS) {
            set_error("graduality job %d: invalid bitmap format (not RGB) or geometry", k);
            return -1;
        }
        GradualDevJob &d = dj[(size_t)k];
        d.format = j.format;
        d.width = j.width;
        d.jump = (kSampleJump % j.width) ? kSampleJump : kSampleJump - 1;
        d.nchunks = 0;
        d.chunk_off = (uint32_t)dc.size();
        d.result_off = (uint32_t)k * 2;
        for (int c = 0; c < j.num_chunks; c++) {
            const uint32_t rows = j.chunk_rows[c];
            if (j.width <= 1 || rows <= 1) {
                trivial[(size_t)k]++;
                continue;
            }
            const std::pair<int, uint32_t> key(j.width, rows);
            std::map<std::pair<int, uint32_t>, uint32_t>::iterator it = tables.find(key);
            if (it == tables.end()) {
                const uint32_t off = (uint32_t)pool.size();
                pool.resize(pool.size() + rows - 1);
                sample_rows(j.width, rows, d.jump, &pool[off]);
                it = tables.insert(std::make_pair(key, off)).first;
            }
            GradualDevChunk ch;
            ch.data = (const uint8_t *)j.chunks[c];
            ch.rows = rows;
            ch.entry_off = it->second;
            dc.push_back(ch);
            d.nchunks++;
        }
    }

    const size_t jobs_b = sizeof(GradualDevJob) * dj.size();
    const size_t chunks_off = (jobs_b + 15) & ~(size_t)15;
    const size_t chunks_b = sizeof(GradualDevChunk) * dc.size();
    const size_t pool_off = (chunks_off + chunks_b + 15) & ~(size_t)15;
    const size_t total = pool_off + sizeof(uint32_t) * pool.size() + 16;
    const size_t res_b = sizeof(long long) * 2 * (size_t)njobs;

    GateHold dev_lock("graduality");
    if (!dev_lock.ok) {
        return -1;
    }
    if (!g_stage.reserve(total) || !g_results.reserve(res_b)) {
        return -1;
    }
    memcpy(g_stage.host, dj.data(), jobs_b);
    if (chunks_b) {
        memcpy(g_stage.host + chunks_off, dc.data(), chunks_b);
    }
    if (!pool.empty()) {
        memcpy(g_stage.host + pool_off, pool.data(), sizeof(uint32_t) * pool.size());
    }
    memset(g_results.host, 0, res_b);
    hipStream_t s = stream();
    bool ok = check(hipMemcpyAsync(g_stage.dev, g_stage.host, total, hipMemcpyHostToDevice, s), "graduality upload") &&
              check(hipMemcpyAsync(g_results.dev, g_results.host, res_b, hipMemcpyHostToDevice, s), "graduality clear");
    if (ok) {
        gradual_launch((const GradualDevJob *)g_stage.dev, njobs, (const GradualDevChunk *)(g_stage.dev + chunks_off),
                       (const uint32_t *)(g_stage.dev + pool_off), (long long *)g_results.dev, s);
        ok = check(hipGetLastError(), "graduality launch") &&
             check(hipMemcpyAsync(g_results.host, g_results.dev, res_b, hipMemcpyDeviceToHost, s), "graduality download") &&
             check(hipStreamSynchronize(s), "graduality sync");
    }
    if (!ok) {
        return -1;
    }
    const long long *res = (const long long *)g_results.host;
    for (int k = 0; k < njobs; k++) {
        SpiceGpuGradualJob &j = jobs[k];
        /* every scored chunk counts three pairs per sample; a chunk too small
         * to score counts one sample of 1.0 (red_bitmap_utils.h:127-131,146) */
        j.score_quarters = res[2 * k] + 4ll * trivial[(size_t)k];
        j.num_samples = 3 * res[2 * k + 1] + trivial[(size_t)k];
        if (j.num_samples == 0) {
            set_error("graduality job %d: no samples", k); /* spice_assert(num_samples), red_worker.c:5949 */
            return -1;
        }
        j.score = ((double)j.score_quarters * 0.25) / (double)j.num_samples;
        j.level = level_of(j.format, j.score);
    }
    return 0;
}

BitmapGradualType spice_gpu_bitmap_graduality_level(int format, int width, int num_chunks,
                                                    const uint8_t *const *chunks, const uint32_t *chunk_rows)
{
    const int bpp = format_bytes(format);
    if (!bpp || width <= 0 || num_chunks <= 0 || num_chunks > SPICE_GPU_GRADUAL_MAX_CHUNKS) {
        set_error("graduality: invalid bitmap format (not RGB) or geometry");
        return BITMAP_GRADUAL_INVALID;
    }
    if (!ensure_init()) {
        return BITMAP_GRADUAL_INVALID;
    }
    SpiceGpuGradualJob job;
    memset(&job, 0, sizeof(job));
    job.format = format;
    job.width = width;
    job.num_chunks = num_chunks;
    std::vector<void *> dev((size_t)num_chunks, nullptr);
    bool ok = true;
    for (int c = 0; c < num_chunks && ok; c++) {
        const size_t bytes = (size_t)chunk_rows[c] * (size_t)width * (size_t)bpp;
        job.chunk_rows[c] = chunk_rows[c];
        if (bytes) {
            ok = check(hipMalloc(&dev[(size_t)c], bytes), "graduality chunk") &&
                 check(hipMemcpy(dev[(size_t)c], chunks[c], bytes, hipMemcpyHostToDevice), "graduality chunk upload");
        }
        job.chunks[c] = dev[(size_t)c];
    }
    if (ok) {
        ok = spice_gpu_graduality_batch(&job, 1) == 0;
    }
    for (int c = 0; c < num_chunks; c++) {
        if (dev[(size_t)c]) {
            (void)hipFree(dev[(size_t)c]);
        }
    }
    return ok ? (BitmapGradualType)job.level : BITMAP_GRADUAL_INVALID;
}

SpiceGpuCodec spice_gpu_choose_codec(const SpiceGpuCodecQuery *q)
{
    static const int is_plt[] = {0, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};   /* BITMAP_FMT_IS_PLT, red_worker.c:511 */
    static const int is_rgb[] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1};   /* BITMAP_FMT_IS_RGB, red_common.h:55 */
    static const int bytes_pp[] = {0, 0, 0, 0, 0, 1, 2, 3, 4, 4, 1}; /* BITMAP_FMP_BYTES_PER_PIXEL, :512 */
    const int MIN_SIZE_TO_COMPRESS = 54, MIN_DIMENSION_TO_QUIC = 3;  /* :6411-6412 */

    if (!q || q->format < 0 || q->format > 10) {
        set_error("choose codec: bad query");
        return SPICE_GPU_CODEC_ERROR;
    }
    const int comp = q->image_compression, fmt = q->format;
    const bool auto_mode = comp == SPICE_IMAGE_COMPRESS_AUTO_LZ || comp == SPICE_IMAGE_COMPRESS_AUTO_GLZ;
    const bool has_graduality = is_rgb[fmt] && fmt != 10; /* BITMAP_FMT_HAS_GRADUALITY, :514-516 */
    /* _stride_is_extra, :5969-5984 */
    bool stride_extra;
    if (is_rgb[fmt]) {
        stride_extra = q->x * (uint32_t)bytes_pp[fmt] < q->stride;
    } else {
        static const int plt_bits[] = {0, 1, 1, 4, 4, 8};
        const uint32_t bytes_width = (q->x * (uint32_t)plt_bits[fmt] + 7) / 8;
        stride_extra = bytes_width < q->stride;
    }
    bool quic = false;

    if (comp == SPICE_IMAGE_COMPRESS_OFF || (uint64_t)q->y * q->stride < (uint64_t)MIN_SIZE_TO_COMPRESS) {
        return SPICE_GPU_CODEC_NONE;
    } else if (comp == SPICE_IMAGE_COMPRESS_QUIC) {
        if (is_plt[fmt]) {
            return SPICE_GPU_CODEC_NONE;
        }
        quic = true;
    } else if (stride_extra || q->chunks_unstable) {
        /* lz handles neither padded strides nor unstable bitmaps */
        if (comp == SPICE_IMAGE_COMPRESS_LZ || comp == SPICE_IMAGE_COMPRESS_GLZ || is_plt[fmt]) {
            return SPICE_GPU_CODEC_NONE;
        }
        quic = true;
    } else if (auto_mode) {
        if (q->x < (uint32_t)MIN_DIMENSION_TO_QUIC || q->y < (uint32_t)MIN_DIMENSION_TO_QUIC) {
            quic = false;
        } else if (q->graduality == BITMAP_GRADUAL_INVALID) {
            if (has_graduality) {
                return SPICE_GPU_CODEC_NEEDS_GRADUALITY;
            }
            quic = false;
        } else {
            quic = q->graduality == BITMAP_GRADUAL_HIGH;
        }
    }

    if (quic) {
        /* picture-like bitmaps go to JPEG when loss is allowed; lz codes the
         * alpha plane, so the stride cannot be padded then (:6470-6479) */
        if (q->can_lossy && q->enable_jpeg && auto_mode &&
            (fmt != SPICE_GPU_BITMAP_FMT_RGBA || !stride_extra)) {
            return SPICE_GPU_CODEC_JPEG;
        }
        return SPICE_GPU_CODEC_QUIC;
    }
    bool glz;
    if (comp == SPICE_IMAGE_COMPRESS_AUTO_GLZ || comp == SPICE_IMAGE_COMPRESS_GLZ) {
        glz = has_graduality && (uint64_t)q->x * q->y < (uint64_t)q->glz_dict_size;
    } else if (comp == SPICE_IMAGE_COMPRESS_AUTO_LZ || comp == SPICE_IMAGE_COMPRESS_LZ) {
        glz = false;
    } else {
        set_error("invalid image compression type %d", comp);
        return SPICE_GPU_CODEC_ERROR;
    }
    /* the global dictionary is used only while it is not frozen (:6499-6507) */
    return glz && !q->glz_frozen ? SPICE_GPU_CODEC_GLZ : SPICE_GPU_CODEC_LZ;
}

} // extern "C"
