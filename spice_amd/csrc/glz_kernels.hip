/*
 * GLZ parse for gfx950: one 256-lane workgroup per image, one image per
 * dictionary.
 *
 * The reference parser (server/glz_encode_tmpl.c:268-491) is a strictly serial
 * walk: at every "stop" it probes a single-slot hash table whose contents
 * depend on all earlier stops.  This kernel keeps that order -- the output is
 * byte-identical -- but evaluates GLZ_WIN consecutive positions per step:
 *
 *   1. gather (all four wavefronts): every lane takes one position of the
 *      window [ip, ip+256): run test, 3-pixel hash, hash-table probe, reference
 *      validation (REF_SEG_IS_VALID, glz_encoder_dictionary_protected.h:176-184),
 *      do_match's distance rules (glz_encode_tmpl.c:147-229) and the first
 *      GLZ_K pixels of the match extension, all as parallel gathers from HBM;
 *      the window's own pixels are staged in LDS;
 *   2. resolve (wavefront 0): visits the real stops in order out of LDS.  A
 *      probe answered in step 1 is still the truth unless a stop *inside the
 *      window* stored the same hash since; ballots over the window's hashes
 *      find that case, and the compare then runs on the LDS tile;
 *   3. matches that outgrow step 1 (the few long ones that cover most pixels)
 *      are extended 512 pixels per iteration; literal and match bytes are
 *      written cooperatively.
 *
 * Hash-table stores are issued by lane 0 only, in stop order, so that memory
 * sees them in the order the reference performs them.
 */
#include <hip/hip_runtime.h>

#include "glz_device.h"
#include "runtime.h"

#define GLZ_WAVE 64
#define GLZ_THREADS 256
#define GLZ_WIN 256                       /* window positions per step */
#define GLZ_K 8                           /* pixels every lane pre-compares against its candidate */
#define GLZ_TILE (GLZ_WIN + GLZ_K + 4)    /* LDS pixels: positions ip-1 .. ip+WIN+K+2 */
#define GLZ_SEG_CACHE 64                  /* segment slots mirrored in LDS */
#define GLZ_UNROLL 8                      /* 64-pixel rows per iteration of the long compares */

#define GLZ_F_ACT 1u
#define GLZ_F_RLE 2u
#define GLZ_F_GOK 4u
#define GLZ_F_MAXED 8u

/* what step 1 leaves in LDS for each window position */
struct GlzPosInfo {
    uint32_t h;
    uint32_t flags;    /* GLZ_F_* | pre-compared length << 8 */
    uint32_t g_ref;
    uint32_t g_seg;
    uint32_t g_image_dist;
    uint32_t g_pix_dist;
};

enum { GLZ_MODE_RGB32 = 0, GLZ_MODE_RGB24 = 1, GLZ_MODE_RGB16 = 2, GLZ_MODE_ALPHA = 3 };

/* Per pixel type constants, server/glz_encode_tmpl.c:33-103,250-256,415-444 */
template <int MODE> struct GlzTraits;
template <> struct GlzTraits<GLZ_MODE_RGB32> {
    static constexpr int PXB = 4, LITB = 3, MIN_REF = 2, MAX_REF = 2, ENC_DIV = 3, LEN_BIAS = 0;
    static constexpr bool GUARD_A = true, GUARD_B = true;
    static constexpr uint32_t MASK = 0x00ffffffu;
};
template <> struct GlzTraits<GLZ_MODE_RGB24> {
    static constexpr int PXB = 3, LITB = 3, MIN_REF = 2, MAX_REF = 2, ENC_DIV = 3, LEN_BIAS = 0;
    static constexpr bool GUARD_A = true, GUARD_B = true;
    static constexpr uint32_t MASK = 0x00ffffffu;
};
template <> struct GlzTraits<GLZ_MODE_RGB16> {
    static constexpr int PXB = 2, LITB = 2, MIN_REF = 2, MAX_REF = 3, ENC_DIV = 2, LEN_BIAS = 1;
    static constexpr bool GUARD_A = true, GUARD_B = false;
    static constexpr uint32_t MASK = 0x00007fffu;
};
template <> struct GlzTraits<GLZ_MODE_ALPHA> {
    static constexpr int PXB = 4, LITB = 1, MIN_REF = 4, MAX_REF = 7, ENC_DIV = 1, LEN_BIAS = 2;
    static constexpr bool GUARD_A = false, GUARD_B = false;
    static constexpr uint32_t MASK = 0x000000ffu;
};

/* raw value of pixel idx: b|g<<8|r<<16(|pad<<24), the 16-bit word, or the pad byte */
template <int MODE>
__device__ __forceinline__ uint32_t glz_load_px(const uint8_t *base, int idx)
{
    if (MODE == GLZ_MODE_RGB32) {
        return ((const uint32_t *)base)[idx];
    } else if (MODE == GLZ_MODE_RGB24) {
        const uint8_t *p = base + (size_t)idx * 3;
        return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
    } else if (MODE == GLZ_MODE_RGB16) {
        return ((const uint16_t *)base)[idx];
    } else {
        return base[(size_t)idx * 4 + 3];
    }
}

#define GLZ_DJB(h, c) ((h) = (((h) << 5) + (h)) ^ (uint32_t)(c))

/* hash of three consecutive (masked) pixels, glz_encode_tmpl.c:41-47,58-64,77-86,110-122 */
template <int MODE>
__device__ __forceinline__ uint32_t glz_hash3(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t h = 5381;
    if (MODE == GLZ_MODE_RGB32 || MODE == GLZ_MODE_RGB24) {
        GLZ_DJB(h, (a >> 16) & 0xff); GLZ_DJB(h, (a >> 8) & 0xff); GLZ_DJB(h, a & 0xff);
        GLZ_DJB(h, (b >> 16) & 0xff); GLZ_DJB(h, (b >> 8) & 0xff); GLZ_DJB(h, b & 0xff);
        GLZ_DJB(h, (c >> 16) & 0xff); GLZ_DJB(h, (c >> 8) & 0xff); GLZ_DJB(h, c & 0xff);
    } else if (MODE == GLZ_MODE_RGB16) {
        GLZ_DJB(h, a & 0xff); GLZ_DJB(h, (a >> 8) & 0x7f);
        GLZ_DJB(h, b & 0xff); GLZ_DJB(h, (b >> 8) & 0x7f);
        GLZ_DJB(h, c & 0xff); GLZ_DJB(h, (c >> 8) & 0x7f);
    } else {
        GLZ_DJB(h, a & 0xff); GLZ_DJB(h, b & 0xff); GLZ_DJB(h, c & 0xff);
    }
    return h & GLZ_HASH_MASK;
}

/* Wave-uniform output cursor.  Every lane holds the same copy. */
struct GlzOut {
    uint8_t *buf;
    uint32_t cap;
    uint32_t op;        /* next byte */
    uint32_t last_copy; /* position of the most recent copy-count byte */
    uint32_t high;      /* furthest position ever reached (overflow detection) */
};

__device__ __forceinline__ void glz_put(GlzOut &o, uint32_t pos, uint32_t b)
{
    if (pos < o.cap) {
        o.buf[pos] = (uint8_t)b;
    }
}

__device__ __forceinline__ void glz_advance(GlzOut &o, uint32_t n)
{
    o.op += n;
    o.high = o.op > o.high ? o.op : o.high;
}

/* glz_encoder.c:89-93 */
__device__ __forceinline__ void glz_put_copy_count(GlzOut &o, uint32_t v, int lane)
{
    if (lane == 0) {
        glz_put(o, o.op, v);
    }
    o.last_copy = o.op;
    glz_advance(o, 1);
}

/* glz_encoder.c:95-108: close the literal run that precedes a match / the end */
__device__ __forceinline__ void glz_close_copy(GlzOut &o, int copy, int lane)
{
    if (copy) {
        if (lane == 0) {
            glz_put(o, o.last_copy, (uint32_t)(copy - 1));
        }
    } else {
        o.op--;
    }
}

/* glz_encode_match_tmpl.c:34-151, size part */
__device__ __forceinline__ int glz_ref_bytes(uint32_t image_dist, uint32_t pix_dist)
{
    int n;
    if (pix_dist < 4096u) {
        n = image_dist < (1u << 6) ? 3 : image_dist < (1u << 14) ? 4 : image_dist < (1u << 22) ? 5 : 6;
    } else {
        n = image_dist == 0 ? 3 : image_dist < (1u << 8) ? 4 : image_dist < (1u << 16) ? 5 : 6;
        if (pix_dist >= 131072u) {
            n++;
        }
    }
    return n;
}

/* glz_encode_match_tmpl.c:31-151, emission part.  len is already biased. */
__device__ void glz_put_match(GlzOut &o, uint32_t image_dist, uint32_t pix_dist, uint32_t len,
                              int lane)
{
    const uint32_t far_px = pix_dist >= 4096u;
    const uint32_t low = (far_px ? 16u : 0u) + (pix_dist & 0x0f);
    uint32_t pos = o.op;
    uint32_t n = 0; /* bytes, counted identically by every lane */

    if (len < 7) {
        if (lane == 0) {
            glz_put(o, pos, (len << 5) + low);
        }
        n = 1;
    } else {
        uint32_t rem = len - 7, nff = rem / 255u, last = rem - nff * 255u;
        if (lane == 0) {
            glz_put(o, pos, (7u << 5) + low);
            glz_put(o, pos + 1 + nff, last);
        }
        for (uint32_t t = (uint32_t)lane; t < nff; t += GLZ_WAVE) {
            glz_put(o, pos + 1 + t, 255u);
        }
        n = 2 + nff;
    }
    if (lane == 0) {
        uint32_t p = pos + n;
        glz_put(o, p++, (pix_dist >> 4) & 255u);
        if (!far_px) {
            if (image_dist < (1u << 6)) {
                glz_put(o, p++, image_dist & 0x3f);
            } else if (image_dist < (1u << 14)) {
                glz_put(o, p++, (1u << 6) + (image_dist & 0x3f));
                glz_put(o, p++, (image_dist >> 6) & 255u);
            } else if (image_dist < (1u << 22)) {
                glz_put(o, p++, (1u << 7) + (image_dist & 0x3f));
                glz_put(o, p++, (image_dist >> 6) & 255u);
                glz_put(o, p++, (image_dist >> 14) & 255u);
            } else {
                glz_put(o, p++, (1u << 7) + (1u << 6) + (image_dist & 0x3f));
                glz_put(o, p++, (image_dist >> 6) & 255u);
                glz_put(o, p++, (image_dist >> 14) & 255u);
                glz_put(o, p++, (image_dist >> 22) & 255u);
            }
        } else {
            uint32_t very_far = pix_dist < 131072u ? 0u : 32u;
            uint32_t hi = (pix_dist >> 12) & 31u;
            if (image_dist == 0) {
                glz_put(o, p++, very_far + hi);
            } else if (image_dist < (1u << 8)) {
                glz_put(o, p++, very_far + (1u << 6) + hi);
                glz_put(o, p++, image_dist & 255u);
            } else if (image_dist < (1u << 16)) {
                glz_put(o, p++, very_far + (1u << 7) + hi);
                glz_put(o, p++, image_dist & 255u);
                glz_put(o, p++, (image_dist >> 8) & 255u);
            } else {
                glz_put(o, p++, very_far + (1u << 7) + (1u << 6) + hi);
                glz_put(o, p++, image_dist & 255u);
                glz_put(o, p++, (image_dist >> 8) & 255u);
                glz_put(o, p++, (image_dist >> 16) & 255u);
            }
            if (very_far) {
                glz_put(o, p++, (pix_dist >> 17) & 255u);
            }
        }
    }
    /* 1 distance byte + the image-distance bytes */
    n += (uint32_t)glz_ref_bytes(image_dist, pix_dist) - 1;
    glz_advance(o, n);
}

/* 256-bit window masks as four words, touched only through compile-time
 * indices so that they stay in registers */
struct GlzMask {
    uint64_t w0, w1, w2, w3;
};
__device__ __forceinline__ bool glz_mask_test(const GlzMask &m, int i)
{
    const int w = i >> 6;
    const uint64_t v = w == 0 ? m.w0 : (w == 1 ? m.w1 : (w == 2 ? m.w2 : m.w3));
    return (v >> (i & 63)) & 1;
}
__device__ __forceinline__ void glz_mask_set(GlzMask &m, int i)
{
    const int w = i >> 6;
    const uint64_t b = 1ull << (i & 63);
    m.w0 |= w == 0 ? b : 0;
    m.w1 |= w == 1 ? b : 0;
    m.w2 |= w == 2 ? b : 0;
    m.w3 |= w == 3 ? b : 0;
}
__device__ __forceinline__ void glz_mask_clear(GlzMask &m)
{
    m.w0 = m.w1 = m.w2 = m.w3 = 0;
}
/* bits of m strictly below position i */
__device__ __forceinline__ GlzMask glz_mask_below(const GlzMask &m, int i)
{
    const int w = i >> 6;
    const uint64_t part = (1ull << (i & 63)) - 1;
    GlzMask r;
    r.w0 = w > 0 ? m.w0 : (m.w0 & part);
    r.w1 = w > 1 ? m.w1 : (w == 1 ? (m.w1 & part) : 0);
    r.w2 = w > 2 ? m.w2 : (w == 2 ? (m.w2 & part) : 0);
    r.w3 = w == 3 ? (m.w3 & part) : 0;
    return r;
}

/* Literal pixels of the window positions in mask[] (bit p of word p/64), in
 * order, with the copy-count bytes that fall between them
 * (glz_encode_tmpl.c:449-462).  Wavefront 0 only; pixels come from the LDS tile. */
template <int MODE>
__device__ __forceinline__ void glz_flush_literals(GlzOut &o, int &copy, const GlzMask &lit,
                                                   const uint32_t *tile, int lane)
{
    typedef GlzTraits<MODE> T;
    const uint64_t mask[4] = {lit.w0, lit.w1, lit.w2, lit.w3};
    const int n0 = __popcll(mask[0]), n1 = __popcll(mask[1]), n2 = __popcll(mask[2]);
    const int n = n0 + n1 + n2 + __popcll(mask[3]);
    if (n == 0) {
        return;
    }
    const int total_ctrl = (copy + n) >> 5;
    int before = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        if ((mask[w] >> lane) & 1) {
            const int k = before + __popcll(mask[w] & ((1ull << lane) - 1));
            const uint32_t raw = tile[w * 64 + lane + 1];
            const uint32_t pos = o.op + (uint32_t)k * T::LITB + (uint32_t)((copy + k) >> 5);
            if (MODE == GLZ_MODE_RGB32 || MODE == GLZ_MODE_RGB24) {
                glz_put(o, pos, raw & 0xff);
                glz_put(o, pos + 1, (raw >> 8) & 0xff);
                glz_put(o, pos + 2, (raw >> 16) & 0xff);
            } else if (MODE == GLZ_MODE_RGB16) {
                glz_put(o, pos, (raw >> 8) & 0xff);
                glz_put(o, pos + 1, raw & 0xff);
            } else {
                glz_put(o, pos, raw & 0xff);
            }
            /* a full run of 32 is followed by the next run's count byte; the last
             * one of this flush may still be patched, so lane 0 owns it (below) */
            if (((copy + k + 1) & 31) == 0 && ((copy + k + 1) >> 5) < total_ctrl) {
                glz_put(o, pos + T::LITB, GLZ_MAX_COPY - 1);
            }
        }
        before += w == 0 ? n0 : (w == 1 ? n1 : n2);
    }
    if (total_ctrl) {
        const int kk = 32 * total_ctrl - copy - 1; /* literal after which the last count byte sits */
        o.last_copy = o.op + (uint32_t)kk * T::LITB + (uint32_t)((copy + kk) >> 5) + T::LITB;
        if (lane == 0) {
            glz_put(o, o.last_copy, GLZ_MAX_COPY - 1);
        }
    }
    glz_advance(o, (uint32_t)n * T::LITB + (uint32_t)total_ctrl);
    copy = (copy + n) & 31;
}

/* Up to 64 literal pixels straight from the window arena (segment heads and
 * tails), glz_encode_tmpl.c:466-487,511-541.  Wavefront 0 only. */
template <int MODE>
__device__ __forceinline__ void glz_literals_from_arena(GlzOut &o, int &copy, const uint8_t *base,
                                                        int first, int count, uint32_t *scratch,
                                                        int lane)
{
    if (count <= 0) {
        return;
    }
    /* reuse the tile layout expected by glz_flush_literals: pixel p at scratch[p + 1] */
    if (lane < count) {
        scratch[lane + 1] = glz_load_px<MODE>(base, first + lane);
    }
    GlzMask mask;
    glz_mask_clear(mask);
    mask.w0 = count >= 64 ? ~0ull : ((1ull << count) - 1);
    glz_flush_literals<MODE>(o, copy, mask, scratch, lane);
}

/* Number of leading positions k in [start, maxn) with src[sidx+k] == ref[ridx+k],
 * plus start; 512 pixels per iteration.  Wavefront 0 only. */
template <int MODE>
__device__ int glz_match_len(const uint8_t *sbase, int sidx, const uint8_t *rbase, int ridx,
                             int start, int maxn, int lane)
{
    typedef GlzTraits<MODE> T;
    int n = start;
    while (n < maxn) {
        uint64_t miss[GLZ_UNROLL];
#pragma unroll
        for (int u = 0; u < GLZ_UNROLL; u++) {
            const int k = n + u * 64 + lane;
            bool eq = false;
            if (k < maxn) {
                eq = ((glz_load_px<MODE>(sbase, sidx + k) ^ glz_load_px<MODE>(rbase, ridx + k)) &
                      T::MASK) == 0;
            }
            miss[u] = __ballot(!eq);
        }
        bool stop = false;
#pragma unroll
        for (int u = 0; u < GLZ_UNROLL; u++) {
            if (!stop) {
                if (miss[u]) {
                    n += __ffsll((unsigned long long)miss[u]) - 1;
                    stop = true;
                } else {
                    n += 64;
                }
            }
        }
        if (stop) {
            break;
        }
    }
    return n < maxn ? n : maxn;
}

/* Number of leading positions k < maxn with src[sidx+k] == x.  Wavefront 0 only. */
template <int MODE>
__device__ int glz_run_len(const uint8_t *sbase, int sidx, uint32_t x, int maxn, int lane)
{
    typedef GlzTraits<MODE> T;
    int n = 0;
    while (n < maxn) {
        uint64_t miss[GLZ_UNROLL];
#pragma unroll
        for (int u = 0; u < GLZ_UNROLL; u++) {
            const int k = n + u * 64 + lane;
            bool eq = false;
            if (k < maxn) {
                eq = ((glz_load_px<MODE>(sbase, sidx + k) ^ x) & T::MASK) == 0;
            }
            miss[u] = __ballot(!eq);
        }
        bool stop = false;
#pragma unroll
        for (int u = 0; u < GLZ_UNROLL; u++) {
            if (!stop) {
                if (miss[u]) {
                    n += __ffsll((unsigned long long)miss[u]) - 1;
                    stop = true;
                } else {
                    n += 64;
                }
            }
        }
        if (stop) {
            break;
        }
    }
    return n < maxn ? n : maxn;
}

/* hash of the three pixels at idx, loaded afresh */
template <int MODE>
__device__ __forceinline__ uint32_t glz_hash_at(const uint8_t *base, int idx)
{
    typedef GlzTraits<MODE> T;
    return glz_hash3<MODE>(glz_load_px<MODE>(base, idx) & T::MASK,
                           glz_load_px<MODE>(base, idx + 1) & T::MASK,
                           glz_load_px<MODE>(base, idx + 2) & T::MASK);
}

/* LDS shared by the workgroup */
struct GlzShared {
    uint32_t tile[GLZ_TILE];
    GlzPosInfo info[GLZ_WIN];
    GlzDevSeg seg_cache[GLZ_SEG_CACHE];
    int next_ip;
};

__shared__ GlzShared glz_sh;

__device__ __forceinline__ GlzDevSeg glz_seg(const GlzDevJob &job, const GlzShared &sh, uint32_t idx)
{
    if (idx < GLZ_SEG_CACHE) {
        return sh.seg_cache[idx];
    }
    return job.segs[idx];
}

/* Matching pixels between window positions i and j < i, read from the LDS tile
 * as far as it reaches; *checked = how many positions the tile could vouch for. */
template <int MODE>
__device__ __forceinline__ int glz_tile_match_len(const uint32_t *tile, int i, int j, int maxn,
                                                  int lane, int *checked)
{
    typedef GlzTraits<MODE> T;
    int lim = GLZ_TILE - (i + 1);
    lim = lim < maxn ? lim : maxn;
    lim = lim < 64 ? lim : 64;
    bool eq = false;
    if (lane < lim) {
        eq = ((tile[i + 1 + lane] ^ tile[j + 1 + lane]) & T::MASK) == 0;
    }
    const uint64_t miss = __ballot(!eq);
    const int n = miss ? __ffsll((unsigned long long)miss) - 1 : 64;
    *checked = lim;
    return n < lim ? n : lim;
}

/* One segment, glz_encode_tmpl.c:268-491.  All 256 lanes call this; GlzOut and
 * copy live in wavefront 0. */
template <int MODE>
__device__ __forceinline__ void glz_parse_seg(const GlzDevJob &job, uint32_t seg_idx, int from,
                                              int copied, GlzOut &o, int tid)
{
    GlzShared &sh = glz_sh;
    typedef GlzTraits<MODE> T;
    const int lane = tid & 63;
    const bool w0 = tid < GLZ_WAVE;
    const GlzDevSeg cs = glz_seg(job, sh, seg_idx);
    const uint8_t *base = job.arena + (size_t)cs.arena_off16 * 16;
    const int npix = (int)cs.nunits;
    const int ip_bound = npix - 2, ip_limit = npix - 6;
    int ip = from;
    int copy = copied;

    if (w0 && copy == 0) {
        glz_put_copy_count(o, GLZ_MAX_COPY - 1, lane);
    }

    while (ip < ip_limit) {
        /* ---------------- step 1: stage the window's pixels, one position per lane */
        for (int t = tid; t < GLZ_TILE; t += GLZ_THREADS) {
            const int pos = ip - 1 + t;
            sh.tile[t] = (pos >= 0 && pos < npix) ? glz_load_px<MODE>(base, pos) : 0u;
        }
        __syncthreads();
        {
            const int p = tid, q = ip + p;
            const bool act = q < ip_limit;
            const uint32_t cm1 = sh.tile[p] & T::MASK, c0 = sh.tile[p + 1] & T::MASK;
            const uint32_t c1 = sh.tile[p + 2] & T::MASK, c2 = sh.tile[p + 3] & T::MASK;
            const bool rle = act && q > 0 && cm1 == c0 && c0 == c1 && c1 == c2;
            GlzPosInfo pi;
            pi.h = glz_hash3<MODE>(c0, c1, c2);
            pi.flags = (act ? GLZ_F_ACT : 0u) | (rle ? GLZ_F_RLE : 0u);
            pi.g_ref = pi.g_seg = pi.g_image_dist = pi.g_pix_dist = 0;
            if (act && !rle) {
                /* hash probe + do_match (glz_encode_tmpl.c:147-241) against the slot's content */
                const uint2 he = job.htab[pi.h];
                const GlzDevSeg rs = glz_seg(job, sh, he.x);
                const bool valid =
                    he.x == seg_idx ||
                    ((rs.flags & GLZ_SEG_USED) && (rs.flags & GLZ_SEG_ALIVE) &&
                     (rs.flags & GLZ_SEG_TYPE_MASK) == (cs.flags & GLZ_SEG_TYPE_MASK) &&
                     rs.pixels_so_far <= cs.pixels_so_far && job.head_psf <= rs.pixels_so_far);
                if (valid && (int64_t)he.y <= (int64_t)rs.nunits - T::MIN_REF) {
                    const uint64_t idist = cs.image_id - rs.image_id;
                    uint64_t pdist;
                    if (idist == 0) {
                        pdist = ((uint64_t)q + cs.pixels_so_far) - ((uint64_t)he.y + rs.pixels_so_far);
                    } else {
                        pdist = (uint64_t)he.y + rs.img_first_delta;
                    }
                    if (pdist != 0 && pdist < 33554432ull && idist <= 16777215ull) {
                        const uint8_t *rbase = job.arena + (size_t)rs.arena_off16 * 16;
                        const int ref = (int)he.y;
                        int maxk = ip_bound - q;
                        const int rmax = (int)rs.nunits - ref;
                        maxk = rmax < maxk ? rmax : maxk;
                        const bool capped = maxk > GLZ_K;
                        maxk = capped ? GLZ_K : maxk;
                        uint32_t r[GLZ_K];
#pragma unroll
                        for (int k = 0; k < GLZ_K; k++) {
                            r[k] = glz_load_px<MODE>(rbase, ref + (k < maxk ? k : 0));
                        }
                        int glen = 0;
                        bool run = true;
#pragma unroll
                        for (int k = 0; k < GLZ_K; k++) {
                            run = run && k < maxk && ((r[k] ^ sh.tile[p + 1 + k]) & T::MASK) == 0;
                            glen += run ? 1 : 0;
                        }
                        if (glen >= T::MIN_REF) {
                            pi.flags |= GLZ_F_GOK | ((uint32_t)glen << 8);
                            if (capped && glen == GLZ_K) {
                                pi.flags |= GLZ_F_MAXED;
                            }
                            pi.g_ref = he.y;
                            pi.g_seg = he.x;
                            pi.g_image_dist = (uint32_t)idist;
                            pi.g_pix_dist = (uint32_t)pdist;
                        }
                    }
                }
            }
            sh.info[p] = pi;
        }
        __syncthreads();

        /* ---------------- step 2: wavefront 0 visits the stops in order */
        if (w0) {
            const uint32_t h0r = sh.info[lane].h, h1r = sh.info[64 + lane].h;
            const uint32_t h2r = sh.info[128 + lane].h, h3r = sh.info[192 + lane].h;
            const uint32_t f0 = sh.info[lane].flags, f1 = sh.info[64 + lane].flags;
            const uint32_t f2 = sh.info[128 + lane].flags, f3 = sh.info[192 + lane].flags;
            GlzMask act_m, rle_m, gok_m;
            act_m.w0 = __ballot(f0 & GLZ_F_ACT); act_m.w1 = __ballot(f1 & GLZ_F_ACT);
            act_m.w2 = __ballot(f2 & GLZ_F_ACT); act_m.w3 = __ballot(f3 & GLZ_F_ACT);
            rle_m.w0 = __ballot(f0 & GLZ_F_RLE); rle_m.w1 = __ballot(f1 & GLZ_F_RLE);
            rle_m.w2 = __ballot(f2 & GLZ_F_RLE); rle_m.w3 = __ballot(f3 & GLZ_F_RLE);
            gok_m.w0 = __ballot(f0 & GLZ_F_GOK); gok_m.w1 = __ballot(f1 & GLZ_F_GOK);
            gok_m.w2 = __ballot(f2 & GLZ_F_GOK); gok_m.w3 = __ballot(f3 & GLZ_F_GOK);
            GlzMask inserted, lit; /* slots stored in this step; positions pending as literals */
            glz_mask_clear(inserted);
            glz_mask_clear(lit);
            int i = 0;
            while (i < GLZ_WIN && glz_mask_test(act_m, i)) {
                const int anchor = ip + i;
                int len = 0;
                uint32_t image_dist = 0, pix_dist = 0;

                if (glz_mask_test(rle_m, i)) {
                    /* glz_encode_tmpl.c:310-335 */
                    const uint32_t x = sh.tile[i + 3];
                    const int maxn = ip_bound - (anchor + 3);
                    int lim = GLZ_TILE - (i + 4);
                    lim = lim < maxn ? lim : maxn;
                    lim = lim < 64 ? lim : 64;
                    bool eq = false;
                    if (lane < lim) {
                        eq = ((sh.tile[i + 4 + lane] ^ x) & T::MASK) == 0;
                    }
                    const uint64_t miss = __ballot(!eq);
                    int n = miss ? __ffsll((unsigned long long)miss) - 1 : 64;
                    n = n < lim ? n : lim;
                    if (n == lim && lim < maxn) {
                        n += glz_run_len<MODE>(base, anchor + 3 + n, x, maxn - n, lane);
                    }
                    len = 3 + n;
                    pix_dist = 1;
                    image_dist = 0;
                } else {
                    const uint32_t hi = sh.info[i].h;
                    /* most recent store to this slot made inside the window, if any */
                    int j = -1;
                    {
                        const GlzMask below = glz_mask_below(inserted, i);
                        const uint64_t m0 = __ballot(h0r == hi) & below.w0;
                        const uint64_t m1 = __ballot(h1r == hi) & below.w1;
                        const uint64_t m2 = __ballot(h2r == hi) & below.w2;
                        const uint64_t m3 = __ballot(h3r == hi) & below.w3;
                        if (m3) {
                            j = 192 + 63 - __clzll((long long)m3);
                        } else if (m2) {
                            j = 128 + 63 - __clzll((long long)m2);
                        } else if (m1) {
                            j = 64 + 63 - __clzll((long long)m1);
                        } else if (m0) {
                            j = 63 - __clzll((long long)m0);
                        }
                    }
                    if (j >= 0) {
                        const int maxn = ip_bound - anchor; /* ref < anchor: its own limit is laxer */
                        int checked = 0;
                        len = glz_tile_match_len<MODE>(sh.tile, i, j, maxn, lane, &checked);
                        if (len == checked && checked < maxn) {
                            len = glz_match_len<MODE>(base, anchor, base, ip + j, len, maxn, lane);
                        }
                        if (len < T::MIN_REF) {
                            len = 0;
                        }
                        image_dist = 0;
                        pix_dist = (uint32_t)(i - j);
                    } else if (glz_mask_test(gok_m, i)) {
                        const GlzPosInfo pi = sh.info[i];
                        len = (int)(pi.flags >> 8);
                        image_dist = pi.g_image_dist;
                        pix_dist = pi.g_pix_dist;
                        if (pi.flags & GLZ_F_MAXED) {
                            const GlzDevSeg rs = glz_seg(job, sh, pi.g_seg);
                            const uint8_t *rbase = job.arena + (size_t)rs.arena_off16 * 16;
                            int maxn = ip_bound - anchor;
                            const int rmax = (int)rs.nunits - (int)pi.g_ref;
                            maxn = rmax < maxn ? rmax : maxn;
                            len = glz_match_len<MODE>(base, anchor, rbase, (int)pi.g_ref, len, maxn, lane);
                        }
                    }
                    if (len > 0 && len <= T::MAX_REF) {
                        /* glz_encode_tmpl.c:243-261 */
                        const int need = glz_ref_bytes(image_dist, pix_dist) / T::ENC_DIV + 1;
                        if (len < need) {
                            len = 0;
                        }
                    }
                    /* glz_encode_tmpl.c:381 */
                    if (lane == 0) {
                        job.htab[hi] = make_uint2(seg_idx, (uint32_t)anchor);
                    }
                    glz_mask_set(inserted, i);
                }

                if (len == 0) {
                    glz_mask_set(lit, i);
                    i++;
                    continue;
                }

                /* ---- match, glz_encode_tmpl.c:387-447 */
                glz_flush_literals<MODE>(o, copy, lit, sh.tile, lane);
                glz_mask_clear(lit);
                if (image_dist == 0) {
                    pix_dist--;
                }
                glz_close_copy(o, copy, lane);
                copy = 0;
                glz_put_match(o, image_dist, pix_dist, (uint32_t)(len - T::LEN_BIAS), lane);
                {
                    const int p0 = anchor + len - 2, p1 = p0 + 1;
                    const bool ins0 = !T::GUARD_A || p0 > anchor;
                    const bool ins1 = !T::GUARD_B || p1 > anchor;
                    const int w0i = p0 - ip, w1i = p1 - ip;
                    uint32_t h0, h1;
                    if (w1i < GLZ_WIN) {
                        /* step 1 hashed every window position whose three pixels exist */
                        h0 = sh.info[w0i].h;
                        h1 = sh.info[w1i].h;
                    } else {
                        h0 = glz_hash_at<MODE>(base, p0);
                        h1 = glz_hash_at<MODE>(base, p1);
                    }
                    if (lane == 0) {
                        if (ins0) {
                            job.htab[h0] = make_uint2(seg_idx, (uint32_t)p0);
                        }
                        if (ins1) {
                            job.htab[h1] = make_uint2(seg_idx, (uint32_t)p1);
                        }
                    }
                    if (w0i < GLZ_WIN && ins0) {
                        glz_mask_set(inserted, w0i);
                    }
                    if (w1i < GLZ_WIN && ins1) {
                        glz_mask_set(inserted, w1i);
                    }
                }
                glz_put_copy_count(o, GLZ_MAX_COPY - 1, lane);
                i += len;
            }
            glz_flush_literals<MODE>(o, copy, lit, sh.tile, lane);
            if (lane == 0) {
                sh.next_ip = ip + i; /* i counts every pixel consumed, also past the window */
            }
        }
        __syncthreads();
        ip = sh.next_ip;
        __syncthreads();
    }

    /* ---- leftovers as literals, glz_encode_tmpl.c:466-487 */
    if (w0) {
        glz_literals_from_arena<MODE>(o, copy, base, ip, npix - ip, sh.tile, lane);
        glz_close_copy(o, copy, lane);
    }
    __syncthreads();
}

/* glz_encode_tmpl.c:498-555 */
template <int MODE>
__device__ __forceinline__ void glz_parse_image(const GlzDevJob &job, const uint32_t *seg_list,
                                                GlzOut &o, int tid)
{
    GlzShared &sh = glz_sh;
    const int lane = tid & 63;
    const bool w0 = tid < GLZ_WAVE;
    uint32_t s = 0;

    /* segments shorter than 4 pixels are copied verbatim */
    for (; s < job.nsegs; s++) {
        const GlzDevSeg cs = glz_seg(job, sh, seg_list[s]);
        if (cs.nunits >= 4) {
            break;
        }
        if (cs.nunits > 0 && w0) {
            const uint8_t *base = job.arena + (size_t)cs.arena_off16 * 16;
            int copy = 0;
            glz_put_copy_count(o, cs.nunits - 1, lane);
            glz_literals_from_arena<MODE>(o, copy, base, 0, (int)cs.nunits, sh.tile, lane);
        }
    }
    bool first = true;
    for (; s < job.nsegs; s++) {
        const uint32_t sid = seg_list[s];
        if (first) {
            /* the first real segment starts with a hash insert and two literals */
            if (w0) {
                const GlzDevSeg cs = glz_seg(job, sh, sid);
                const uint8_t *base = job.arena + (size_t)cs.arena_off16 * 16;
                int copy = 0;
                glz_put_copy_count(o, GLZ_MAX_COPY - 1, lane);
                if (lane == 0) {
                    job.htab[glz_hash_at<MODE>(base, 0)] = make_uint2(sid, 0);
                }
                glz_literals_from_arena<MODE>(o, copy, base, 0, 2, sh.tile, lane);
            }
            __syncthreads();
        }
        glz_parse_seg<MODE>(job, sid, first ? 2 : 0, first ? 2 : 0, o, tid);
        first = false;
    }
}

/* One workgroup per job: apply the window-table updates, write the header
 * (glz_encoder.c:264-276), parse, report the size. */
template <int TYPE>
__global__ void __launch_bounds__(GLZ_THREADS)
glz_parse_kernel(const GlzDevJob *__restrict__ jobs, const uint32_t *__restrict__ u32_pool,
                 const GlzDevSegUpdate *__restrict__ upd_pool)
{
    GlzShared &sh = glz_sh;
    const GlzDevJob job = jobs[blockIdx.x];
    if ((int)job.type != TYPE) {
        return; /* another instantiation of this kernel handles that pixel type */
    }
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;

    for (uint32_t u = (uint32_t)tid; u < job.nupd; u += GLZ_THREADS) {
        const GlzDevSegUpdate up = upd_pool[job.upd_off + u];
        job.segs[up.index] = up.seg;
    }
    __syncthreads();
    if (tid < GLZ_SEG_CACHE) {
        sh.seg_cache[tid] = job.segs[tid];
    }
    __syncthreads();

    GlzOut o;
    o.buf = job.out;
    o.cap = job.out_cap;
    o.op = 0;
    o.last_copy = 0;
    o.high = 0;

    if (tid < GLZ_HEADER_BYTES) {
        uint32_t b;
        const uint32_t id_hi = (uint32_t)(job.image_id >> 32);
        const uint32_t id_lo = (uint32_t)(job.image_id & 0xffffffu); /* glz_encoder.c:86 */
        if (lane < 4) {
            b = (0x20205a4cu >> (8 * (3 - lane))) & 0xff;
        } else if (lane < 8) {
            b = (0x00010001u >> (8 * (7 - lane))) & 0xff;
        } else if (lane == 8) {
            b = (job.type & 0x0f) | (job.top_down ? 0x10u : 0u);
        } else if (lane < 13) {
            b = (job.width >> (8 * (12 - lane))) & 0xff;
        } else if (lane < 17) {
            b = (job.height >> (8 * (16 - lane))) & 0xff;
        } else if (lane < 21) {
            b = (job.stride >> (8 * (20 - lane))) & 0xff;
        } else if (lane < 25) {
            b = (id_hi >> (8 * (24 - lane))) & 0xff;
        } else if (lane < 29) {
            b = (id_lo >> (8 * (28 - lane))) & 0xff;
        } else {
            b = (job.win_head_dist >> (8 * (32 - lane))) & 0xff;
        }
        glz_put(o, (uint32_t)lane, b);
    }
    glz_advance(o, GLZ_HEADER_BYTES);

    const uint32_t *seg_list = u32_pool + job.seg_list_off;
    if (TYPE == 6) {
        glz_parse_image<GLZ_MODE_RGB16>(job, seg_list, o, tid);
    } else if (TYPE == 7) {
        glz_parse_image<GLZ_MODE_RGB24>(job, seg_list, o, tid);
    } else {
        glz_parse_image<GLZ_MODE_RGB32>(job, seg_list, o, tid);
        if (TYPE == 9) {
            __syncthreads();
            glz_parse_image<GLZ_MODE_ALPHA>(job, seg_list, o, tid);
        }
    }
    if (tid == 0) {
        *job.result = o.high > o.cap ? GLZ_DEV_ERR_NO_SPACE : (int32_t)o.op;
    }
}

/* Rows into the window arena: descs[blockIdx.y], 16 bytes per lane. */
__global__ void __launch_bounds__(256)
glz_copy_kernel(const GlzCopyDesc *descs)
{
    const GlzCopyDesc d = descs[blockIdx.y];
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if ((((uintptr_t)d.src | (uintptr_t)d.dst) & 15) == 0) {
        const uint64_t n16 = d.bytes >> 4;
        const uint4 *s = (const uint4 *)d.src;
        uint4 *w = (uint4 *)d.dst;
        for (uint64_t k = t; k < n16; k += stride) {
            w[k] = s[k];
        }
        for (uint64_t k = (n16 << 4) + t; k < d.bytes; k += stride) {
            d.dst[k] = d.src[k];
        }
    } else {
        for (uint64_t k = t; k < d.bytes; k += stride) {
            d.dst[k] = d.src[k];
        }
    }
}

/* type_mask: bit t set when some job has LzImageType t (6..9) */
void glz_launch_parse(const GlzDevJob *jobs, int njobs, unsigned type_mask, const uint32_t *u32_pool,
                      const GlzDevSegUpdate *upd_pool, hipStream_t stream)
{
    const dim3 grid((unsigned)njobs), block(GLZ_THREADS);
    const int t = spicegpu::timing_begin("glz_parse_kernel", stream);
    if (type_mask & (1u << 8)) {
        hipLaunchKernelGGL(glz_parse_kernel<8>, grid, block, 0, stream, jobs, u32_pool, upd_pool);
    }
    if (type_mask & (1u << 9)) {
        hipLaunchKernelGGL(glz_parse_kernel<9>, grid, block, 0, stream, jobs, u32_pool, upd_pool);
    }
    if (type_mask & (1u << 7)) {
        hipLaunchKernelGGL(glz_parse_kernel<7>, grid, block, 0, stream, jobs, u32_pool, upd_pool);
    }
    if (type_mask & (1u << 6)) {
        hipLaunchKernelGGL(glz_parse_kernel<6>, grid, block, 0, stream, jobs, u32_pool, upd_pool);
    }
    spicegpu::timing_end(t, stream);
}

void glz_launch_copy(const GlzCopyDesc *descs, int ndescs, unsigned blocks_per_desc,
                     hipStream_t stream)
{
    const int t = spicegpu::timing_begin("glz_copy_kernel", stream);
    hipLaunchKernelGGL(glz_copy_kernel, dim3(blocks_per_desc, (unsigned)ndescs), dim3(256), 0,
                       stream, descs);
    spicegpu::timing_end(t, stream);
}
