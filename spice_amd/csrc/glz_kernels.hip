/*
 * GLZ parse for gfx950: one wavefront per image, one image per dictionary.
 *
 * The reference parser (server/glz_encode_tmpl.c:268-491) is a strictly serial
 * walk: at every "stop" it probes a single-slot hash table whose contents
 * depend on all earlier stops.  This kernel keeps that order -- the output is
 * byte-identical -- but evaluates 64 consecutive positions per step:
 *
 *   1. every lane takes one position of the window [ip, ip+64): run test,
 *      3-pixel hash, hash-table probe, reference validation
 *      (REF_SEG_IS_VALID, glz_encoder_dictionary_protected.h:176-184) and the
 *      first pixel compares of do_match (glz_encode_tmpl.c:147-229), all as
 *      parallel gathers;
 *   2. a wave-uniform resolver then visits the real stops in order.  A probe
 *      answered in step 1 is still the truth unless a stop *inside the
 *      window* stored the same hash since; one ballot finds that case;
 *   3. match extension, run extension and the literal/match byte emission are
 *      cooperative (64 pixels compared or written per instruction).
 *
 * Hash-table stores are issued by lane 0 only, in stop order, so that memory
 * sees them in the order the reference performs them.
 */
#include <hip/hip_runtime.h>

#include "glz_device.h"
#include "runtime.h"

#define GLZ_WAVE 64

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

/* Writes the literal pixels of the lanes in `mask` (window order) and the
 * copy-count bytes that fall between them, glz_encode_tmpl.c:449-462. */
template <int MODE>
__device__ __forceinline__ void glz_flush_literals(GlzOut &o, int &copy, uint64_t mask,
                                                   uint32_t raw, int lane)
{
    typedef GlzTraits<MODE> T;
    const int n = __popcll(mask);
    if (n == 0) {
        return;
    }
    if ((mask >> lane) & 1) {
        const int k = __popcll(mask & ((1ull << lane) - 1));
        uint32_t pos = o.op + (uint32_t)k * T::LITB + (uint32_t)((copy + k) >> 5);
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
        if (((copy + k + 1) & 31) == 0 && ((copy + k + 1) >> 5) < ((copy + n) >> 5)) {
            glz_put(o, pos + T::LITB, GLZ_MAX_COPY - 1);
        }
    }
    const int nctrl = (copy + n) >> 5;
    if (nctrl) {
        const int kk = 32 * nctrl - copy - 1; /* literal after which the last count byte sits */
        o.last_copy = o.op + (uint32_t)kk * T::LITB + (uint32_t)((copy + kk) >> 5) + T::LITB;
        if (lane == 0) {
            glz_put(o, o.last_copy, GLZ_MAX_COPY - 1);
        }
    }
    glz_advance(o, (uint32_t)n * T::LITB + (uint32_t)nctrl);
    copy = (copy + n) & 31;
}

/* Number of leading positions k < maxn with src[sidx+k] == ref[ridx+k]. */
template <int MODE>
__device__ int glz_match_len(const uint8_t *sbase, int sidx, const uint8_t *rbase, int ridx,
                             int start, int maxn, int lane)
{
    typedef GlzTraits<MODE> T;
    int n = start;
    while (n < maxn) {
        const int k = n + lane;
        bool eq = false;
        if (k < maxn) {
            eq = ((glz_load_px<MODE>(sbase, sidx + k) ^ glz_load_px<MODE>(rbase, ridx + k)) &
                  T::MASK) == 0;
        }
        const uint64_t miss = __ballot(!eq);
        if (miss) {
            n += __ffsll((unsigned long long)miss) - 1;
            break;
        }
        n += GLZ_WAVE;
    }
    return n < maxn ? n : maxn;
}

/* Number of leading positions k < maxn with src[sidx+k] == x. */
template <int MODE>
__device__ int glz_run_len(const uint8_t *sbase, int sidx, uint32_t x, int maxn, int lane)
{
    typedef GlzTraits<MODE> T;
    int n = 0;
    while (n < maxn) {
        const int k = n + lane;
        bool eq = false;
        if (k < maxn) {
            eq = ((glz_load_px<MODE>(sbase, sidx + k) ^ x) & T::MASK) == 0;
        }
        const uint64_t miss = __ballot(!eq);
        if (miss) {
            n += __ffsll((unsigned long long)miss) - 1;
            break;
        }
        n += GLZ_WAVE;
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

/* One segment, glz_encode_tmpl.c:268-491. */
template <int MODE>
__device__ void glz_parse_seg(const GlzDevJob &job, uint32_t seg_idx, int from, int copied,
                              GlzOut &o, int lane)
{
    typedef GlzTraits<MODE> T;
    const GlzDevSeg cs = job.segs[seg_idx];
    const uint8_t *base = job.arena + (size_t)cs.arena_off16 * 16;
    const int npix = (int)cs.nunits;
    const int ip_bound = npix - 2, ip_limit = npix - 6;
    int ip = from;
    int copy = copied;

    if (copy == 0) {
        glz_put_copy_count(o, GLZ_MAX_COPY - 1, lane);
    }

    while (ip < ip_limit) {
        /* ---------------- step 1: the window, one position per lane */
        const int q = ip + lane;
        const bool act = q < ip_limit;
        uint32_t raw0 = 0, ext = 0;
        if (q < npix) {
            raw0 = glz_load_px<MODE>(base, q);
        }
        {
            /* pixels just outside the window that its edge lanes need */
            int e = lane == 0 ? ip + 64 : lane == 1 ? ip + 65 : lane == 2 ? ip - 1 : -1;
            if (lane < 3 && e >= 0 && e < npix) {
                ext = glz_load_px<MODE>(base, e);
            }
        }
        const uint32_t c0 = raw0 & T::MASK;
        const uint32_t d1 = __shfl_down(c0, 1), d2 = __shfl_down(c0, 2), u1 = __shfl_up(c0, 1);
        const uint32_t e0 = __shfl(ext, 0) & T::MASK, e1 = __shfl(ext, 1) & T::MASK;
        const uint32_t e2 = __shfl(ext, 2) & T::MASK;
        const uint32_t c1 = lane < 63 ? d1 : e0;
        const uint32_t c2 = lane < 62 ? d2 : (lane == 62 ? e0 : e1);
        const uint32_t cm1 = lane > 0 ? u1 : e2;

        const bool rle = act && q > 0 && cm1 == c0 && c0 == c1 && c1 == c2;
        const uint32_t h = glz_hash3<MODE>(c0, c1, c2);

        /* hash probe + do_match up to (not including) the extension loop */
        bool gok = false;
        uint32_t g_image_dist = 0, g_pix_dist = 0, g_ref = 0, g_seg = 0;
        if (act && !rle) {
            const uint2 he = job.htab[h];
            const GlzDevSeg rs = job.segs[he.x];
            const bool valid =
                he.x == seg_idx ||
                ((rs.flags & GLZ_SEG_USED) && (rs.flags & GLZ_SEG_ALIVE) &&
                 (rs.flags & GLZ_SEG_TYPE_MASK) == (cs.flags & GLZ_SEG_TYPE_MASK) &&
                 rs.pixels_so_far <= cs.pixels_so_far && job.head_psf <= rs.pixels_so_far);
            const int ref = (int)he.y;
            if (valid && (int64_t)he.y <= (int64_t)rs.nunits - T::MIN_REF) {
                const uint8_t *rbase = job.arena + (size_t)rs.arena_off16 * 16;
                bool same = ((glz_load_px<MODE>(rbase, ref) & T::MASK) == c0) &&
                            ((glz_load_px<MODE>(rbase, ref + 1) & T::MASK) == c1);
                if (T::MIN_REF == 4) {
                    same = same && ((glz_load_px<MODE>(rbase, ref + 2) & T::MASK) == c2) &&
                           (((glz_load_px<MODE>(rbase, ref + 3) ^
                              glz_load_px<MODE>(base, q + 3)) & T::MASK) == 0);
                }
                if (same) {
                    const uint64_t idist = cs.image_id - rs.image_id;
                    uint64_t pdist;
                    if (idist == 0) {
                        pdist = ((uint64_t)q + cs.pixels_so_far) - ((uint64_t)he.y + rs.pixels_so_far);
                    } else {
                        pdist = (uint64_t)he.y + rs.img_first_delta;
                    }
                    if (pdist != 0 && pdist < 33554432ull && idist <= 16777215ull) {
                        gok = true;
                        g_image_dist = (uint32_t)idist;
                        g_pix_dist = (uint32_t)pdist;
                        g_ref = he.y;
                        g_seg = he.x;
                    }
                }
            }
        }
        const uint64_t act_mask = __ballot(act);
        const uint64_t rle_mask = __ballot(rle);
        const uint64_t gok_mask = __ballot(gok);

        /* ---------------- step 2: visit the stops in order */
        uint64_t inserted = 0; /* window lanes whose hash slot was stored in this step */
        uint64_t lit_mask = 0; /* window lanes pending emission as literals */
        int i = 0;
        while (i < 64 && ((act_mask >> i) & 1)) {
            const int anchor = ip + i;
            int len = 0;
            uint32_t image_dist = 0, pix_dist = 0;

            if ((rle_mask >> i) & 1) {
                /* glz_encode_tmpl.c:310-335 */
                const uint32_t x = __shfl(c2, i);
                len = 3 + glz_run_len<MODE>(base, anchor + 3, x, ip_bound - (anchor + 3), lane);
                pix_dist = 1;
                image_dist = 0;
            } else {
                const uint32_t hi = __shfl(h, i);
                const uint64_t same_h =
                    __ballot(h == hi) & inserted & ((1ull << i) - 1);
                if (same_h) {
                    /* the slot was overwritten by a stop inside this window */
                    const int j = 63 - __clzll((long long)same_h);
                    const int ref = ip + j;
                    const int maxn = ip_bound - anchor; /* ref < anchor, so its limit is laxer */
                    len = glz_match_len<MODE>(base, anchor, base, ref, 0, maxn, lane);
                    if (len < T::MIN_REF) {
                        len = 0;
                    }
                    image_dist = 0;
                    pix_dist = (uint32_t)(i - j);
                } else if ((gok_mask >> i) & 1) {
                    const uint32_t rseg = __shfl(g_seg, i);
                    const int ref = (int)__shfl(g_ref, i);
                    const GlzDevSeg rs = job.segs[rseg];
                    const uint8_t *rbase = job.arena + (size_t)rs.arena_off16 * 16;
                    int maxn = ip_bound - anchor;
                    const int rmax = (int)rs.nunits - ref;
                    maxn = rmax < maxn ? rmax : maxn;
                    len = glz_match_len<MODE>(base, anchor, rbase, ref, T::MIN_REF, maxn, lane);
                    image_dist = __shfl(g_image_dist, i);
                    pix_dist = __shfl(g_pix_dist, i);
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
                inserted |= 1ull << i;
            }

            if (len == 0) {
                lit_mask |= 1ull << i;
                i++;
                continue;
            }

            /* ---- match, glz_encode_tmpl.c:387-447 */
            glz_flush_literals<MODE>(o, copy, lit_mask, raw0, lane);
            lit_mask = 0;
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
                uint32_t h0, h1;
                if (p1 - ip < 64 && ((act_mask >> (p1 - ip)) & 1)) {
                    h0 = __shfl(h, p0 - ip);
                    h1 = __shfl(h, p1 - ip);
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
                if (p0 - ip < 64 && ins0) {
                    inserted |= 1ull << (p0 - ip);
                }
                if (p1 - ip < 64 && ins1) {
                    inserted |= 1ull << (p1 - ip);
                }
            }
            glz_put_copy_count(o, GLZ_MAX_COPY - 1, lane);
            i += len;
        }
        glz_flush_literals<MODE>(o, copy, lit_mask, raw0, lane);
        ip += i; /* i counts every pixel consumed, including a match that ran past the window */
    }

    /* ---- leftovers as literals, glz_encode_tmpl.c:466-487 */
    {
        const int left = npix - ip; /* positions ip .. npix-1 */
        uint32_t raw = 0;
        uint64_t mask = 0;
        if (left > 0) {
            if (lane < left) {
                raw = glz_load_px<MODE>(base, ip + lane);
            }
            mask = left >= 64 ? ~0ull : ((1ull << left) - 1);
        }
        glz_flush_literals<MODE>(o, copy, mask, raw, lane);
        glz_close_copy(o, copy, lane);
    }
}

/* glz_encode_tmpl.c:498-555 */
template <int MODE>
__device__ void glz_parse_image(const GlzDevJob &job, const uint32_t *seg_list, GlzOut &o, int lane)
{
    typedef GlzTraits<MODE> T;
    uint32_t s = 0;

    /* segments shorter than 4 pixels are copied verbatim */
    for (; s < job.nsegs; s++) {
        const GlzDevSeg cs = job.segs[seg_list[s]];
        if (cs.nunits >= 4) {
            break;
        }
        if (cs.nunits > 0) {
            const uint8_t *base = job.arena + (size_t)cs.arena_off16 * 16;
            int copy = 0;
            uint32_t raw = 0;
            glz_put_copy_count(o, cs.nunits - 1, lane);
            if (lane < (int)cs.nunits) {
                raw = glz_load_px<MODE>(base, lane);
            }
            glz_flush_literals<MODE>(o, copy, (1ull << cs.nunits) - 1, raw, lane);
        }
    }
    if (s == job.nsegs) {
        return;
    }
    {
        const uint32_t sid = seg_list[s];
        const GlzDevSeg cs = job.segs[sid];
        const uint8_t *base = job.arena + (size_t)cs.arena_off16 * 16;
        int copy = 0;
        uint32_t raw = 0;
        glz_put_copy_count(o, GLZ_MAX_COPY - 1, lane);
        if (lane == 0) {
            job.htab[glz_hash_at<MODE>(base, 0)] = make_uint2(sid, 0);
        }
        if (lane < 2) {
            raw = glz_load_px<MODE>(base, lane);
        }
        glz_flush_literals<MODE>(o, copy, 3ull, raw, lane);
        glz_parse_seg<MODE>(job, sid, 2, 2, o, lane);
    }
    for (s++; s < job.nsegs; s++) {
        glz_parse_seg<MODE>(job, seg_list[s], 0, 0, o, lane);
    }
    (void)sizeof(T);
}

/* One wavefront per job: apply the window-table updates, write the header
 * (glz_encoder.c:264-276), parse, report the size. */
__global__ void __launch_bounds__(GLZ_WAVE)
glz_parse_kernel(const GlzDevJob *jobs, const uint32_t *u32_pool, const GlzDevSegUpdate *upd_pool)
{
    const GlzDevJob job = jobs[blockIdx.x];
    const int lane = (int)threadIdx.x;

    for (uint32_t u = (uint32_t)lane; u < job.nupd; u += GLZ_WAVE) {
        const GlzDevSegUpdate up = upd_pool[job.upd_off + u];
        job.segs[up.index] = up.seg;
    }
    __syncthreads();

    GlzOut o;
    o.buf = job.out;
    o.cap = job.out_cap;
    o.op = 0;
    o.last_copy = 0;
    o.high = 0;

    if (lane < GLZ_HEADER_BYTES) {
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
    int err = 0;
    switch (job.type) {
    case 6:
        glz_parse_image<GLZ_MODE_RGB16>(job, seg_list, o, lane);
        break;
    case 7:
        glz_parse_image<GLZ_MODE_RGB24>(job, seg_list, o, lane);
        break;
    case 8:
        glz_parse_image<GLZ_MODE_RGB32>(job, seg_list, o, lane);
        break;
    case 9:
        glz_parse_image<GLZ_MODE_RGB32>(job, seg_list, o, lane);
        glz_parse_image<GLZ_MODE_ALPHA>(job, seg_list, o, lane);
        break;
    default:
        err = GLZ_DEV_ERR_BAD_TYPE;
        break;
    }
    if (lane == 0) {
        *job.result = err ? err : (o.high > o.cap ? GLZ_DEV_ERR_NO_SPACE : (int32_t)o.op);
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

void glz_launch_parse(const GlzDevJob *jobs, int njobs, const uint32_t *u32_pool,
                      const GlzDevSegUpdate *upd_pool, hipStream_t stream)
{
    const int t = spicegpu::timing_begin("glz_parse_kernel", stream);
    hipLaunchKernelGGL(glz_parse_kernel, dim3((unsigned)njobs), dim3(GLZ_WAVE), 0, stream, jobs,
                       u32_pool, upd_pool);
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
