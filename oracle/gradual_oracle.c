/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's bitmap
 * graduality score and of red_compress_image's codec choice.  Not linked into
 * the product; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline use it.
 *
 * Follows server/red_bitmap_utils.h:43-147 (pixelcmp, pixels_square_score,
 * compute_lines_gradual_score), server/red_worker.c:5904-5967
 * (_get_bitmap_graduality_level) and :6411-6525 (red_compress_image), in the
 * same order and with the same double arithmetic.  The reference holds no
 * fixture for either: parity unpinned.
 */
#include <stdint.h>
#include <stdlib.h>

#define SAMPLE_JUMP 15

static const double pair_score[3] = {0.5, 1.0, -0.25};

typedef struct {
    int r, g, b;
} Px;

static Px load_px(const uint8_t *lines, int format, size_t idx)
{
    Px p;
    if (format == 6) {
        unsigned v = (unsigned)lines[idx * 2] | ((unsigned)lines[idx * 2 + 1] << 8);
        p.r = (v >> 10) & 0x1f;
        p.g = (v >> 5) & 0x1f;
        p.b = v & 0x1f;
    } else {
        size_t bpp = format == 7 ? 3 : 4;
        p.b = lines[idx * bpp];
        p.g = lines[idx * bpp + 1];
        p.r = lines[idx * bpp + 2];
    }
    return p;
}

/* red_bitmap_utils.h:65-94 */
static int pixelcmp(Px a, Px b, int th)
{
    int diff = abs(a.r - b.r), equal;
    if (diff >= th) {
        return 1;
    }
    equal = !diff;
    diff = abs(a.g - b.g);
    if (diff >= th) {
        return 1;
    }
    equal = equal && !diff;
    diff = abs(a.b - b.b);
    if (diff >= th) {
        return 1;
    }
    equal = equal && !diff;
    return equal ? 0 : 2;
}

/* red_bitmap_utils.h:96-117 */
static double square_score(const uint8_t *lines, int format, size_t top, size_t bottom, int th)
{
    double ret = 0.0;
    int all_ident = 1, c;
    Px p = load_px(lines, format, top);
    c = pixelcmp(p, load_px(lines, format, top + 1), th);
    all_ident = all_ident && !c;
    ret += pair_score[c];
    c = pixelcmp(p, load_px(lines, format, bottom), th);
    all_ident = all_ident && !c;
    ret += pair_score[c];
    c = pixelcmp(p, load_px(lines, format, bottom + 1), th);
    all_ident = all_ident && !c;
    ret += pair_score[c];
    if (all_ident) {
        ret -= pair_score[0] * 3;
    }
    return ret;
}

/* red_bitmap_utils.h:119-147 */
static void lines_score(const uint8_t *lines, int format, int width, int num_lines, double *o_sum,
                        int *o_samples)
{
    int jump = (SAMPLE_JUMP % width) ? SAMPLE_JUMP : SAMPLE_JUMP - 1;
    int th = format == 6 ? 8 : 60;
    int64_t cur = width / 2;
    int64_t last_line = (int64_t)(num_lines - 1) * width;

    if (width <= 1 || num_lines <= 1) {
        *o_samples = 1;
        *o_sum = 1.0;
        return;
    }
    *o_sum = 0;
    *o_samples = 0;
    while (cur < last_line) {
        if ((cur + 1) % width == 0) {
            cur--;
        }
        *o_sum += square_score(lines, format, (size_t)cur, (size_t)(cur + width), th);
        (*o_samples)++;
        cur += jump;
    }
    *o_samples *= 3;
}

/* red_worker.c:5913-5967.  Returns BitmapGradualType (2 LOW, 3 MEDIUM, 4 HIGH). */
int grado_level(int format, int width, int num_chunks, const uint8_t *const *chunks,
                const uint32_t *chunk_rows, double *o_score, int64_t *o_samples)
{
    double score = 0.0, chunk_score = 0.0;
    int64_t samples = 0;
    int chunk_samples = 0, i;

    for (i = 0; i < num_chunks; i++) {
        lines_score(chunks[i], format, width, (int)chunk_rows[i], &chunk_score, &chunk_samples);
        score += chunk_score;
        samples += chunk_samples;
    }
    score /= (double)samples;
    if (o_score) {
        *o_score = score;
    }
    if (o_samples) {
        *o_samples = samples;
    }
    if (format == 6) {
        if (score < 0) {
            return 4;
        }
    } else if (score < -0.03) {
        return 4;
    }
    return score < 0.002 ? 3 : 2;
}

/* red_worker.c:6411-6525; the numbering of the result follows SpiceGpuCodec
 * (0 none, 1 QUIC, 2 JPEG, 3 GLZ, 4 LZ, -1 graduality needed, -2 error). */
int grado_choose(int comp, int fmt, uint32_t x, uint32_t y, uint32_t stride, int unstable, int can_lossy,
                 int enable_jpeg, int graduality, uint32_t glz_dict_size, int glz_frozen)
{
    static const int is_plt[] = {0, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
    static const int is_rgb[] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1};
    static const int bpp[] = {0, 0, 0, 0, 0, 1, 2, 3, 4, 4, 1};
    int quic = 0, extra, has_grad, glz;
    enum { OFF = 1, AUTO_GLZ, AUTO_LZ, QUIC, GLZ, LZ };

    if (fmt < 1 || fmt > 10) {
        return -2;
    }
    has_grad = is_rgb[fmt] && fmt != 10;
    if (is_rgb[fmt]) {
        extra = x * (uint32_t)bpp[fmt] < stride;
    } else if (fmt == 5) {
        extra = x < stride;
    } else if (fmt == 3 || fmt == 4) {
        extra = ((x + 1) & ~1u) >> 1 < stride;
    } else {
        extra = ((x + 7) & ~7u) >> 3 < stride;
    }

    if (comp == OFF || (uint64_t)y * stride < 54) {
        return 0;
    } else if (comp == QUIC) {
        if (is_plt[fmt]) {
            return 0;
        }
        quic = 1;
    } else {
        if (extra || unstable) {
            if (comp == LZ || comp == GLZ || is_plt[fmt]) {
                return 0;
            }
            quic = 1;
        } else {
            if (comp == AUTO_LZ || comp == AUTO_GLZ) {
                if (x < 3 || y < 3) {
                    quic = 0;
                } else if (graduality == 0) {
                    if (has_grad) {
                        return -1;
                    }
                    quic = 0;
                } else {
                    quic = graduality == 4;
                }
            } else {
                quic = 0;
            }
        }
    }
    if (quic) {
        if (can_lossy && enable_jpeg && (comp == AUTO_LZ || comp == AUTO_GLZ)) {
            if (fmt != 9 || !extra) {
                return 2;
            }
        }
        return 1;
    }
    if (comp == AUTO_GLZ || comp == GLZ) {
        glz = has_grad && (uint64_t)x * y < glz_dict_size;
    } else if (comp == AUTO_LZ || comp == LZ) {
        glz = 0;
    } else {
        return -2;
    }
    if (glz && glz_frozen) {
        glz = 0;
    }
    return glz ? 3 : 4;
}
