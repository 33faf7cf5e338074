/*
 * The phases of the GLZ parse (scan, resolve, emit; see glz_kernels.hip) as device
 * functions.  Included by glz_kernels.hip, whose kernel runs them one after the
 * other on one window in 20 KB of LDS, eight workgroups per CU, and -- with
 * GLZ_LONE defined -- by glz_lone.hip, whose kernel runs them side by side on three
 * windows for a GPU that holds fewer sessions than it has CUs.
 */
#define GLZ_WAVE 64
#define GLZ_THREADS 256
#define GLZ_WIN 256                       /* window positions per step */
#define GLZ_K 8                           /* pixels every lane pre-compares against a candidate; above every type's MAX_REF */
#define GLZ_TILE (GLZ_WIN + GLZ_K + 4)    /* LDS pixels: positions ip-1 .. ip+WIN+K+2 */
#ifdef GLZ_STAMPS
#define GLZ_SEG_CACHE 48                  /* the diagnostic build's counters take the place of eight slots */
#else
#define GLZ_SEG_CACHE 56                  /* segment slots mirrored in LDS (what 20 KB per workgroup leave) */
#endif
#define GLZ_KEYS 256                      /* rows of the same-hash bitmap */
/* Word v (32 positions) of a key's row.  Rows are eight words, so with the words in their
 * natural places every lane of a wavefront that reads "its row's word v" would land in one
 * of four LDS banks (v, v + 8, v + 16, v + 24): a 16-way conflict on each of the kernel's
 * hottest LDS accesses.  The words of a row are stored rotated by key / 4, which spreads
 * the same access over all 32 banks. */
#define GLZ_BM(key, v) ((key) * 8u + (((v) + ((key) >> 2)) & 7u))
#define GLZ_NO_HASH 0xffffffffu

/* first scan word of a position */
#define GLZ_A_ACT 0x1u                    /* a stop may happen here (before ip_limit) */
#define GLZ_A_RLE 0x2u                    /* the run test succeeds */
#define GLZ_A_P1V 0x8u                    /* an earlier window position has the same hash */
#define GLZ_A_P2V 0x10u                   /* ... and a second one */
#define GLZ_A_P1_SHIFT 5                  /* 8 bits: nearest such position */
#define GLZ_A_LEN1_SHIFT 13               /* 5 bits: accepted length against it (0 = rejected) */
#define GLZ_A_MAX1 0x40000u               /* ... compare stopped at GLZ_K, not at a difference */
#define GLZ_A_GLEN_SHIFT 19               /* 5 bits: accepted length against the table's entry */
#define GLZ_A_GMAX 0x1000000u
#define GLZ_A_P1SAME 0x2000000u           /* that position lies in the same quarter (64 positions) */
#define GLZ_A_P2SAME 0x4000000u
#define GLZ_A_MORE 0x8000000u             /* a third earlier window position has the same hash */
/* second scan word */
#define GLZ_B_LEN2_SHIFT 8
#define GLZ_B_MAX2 0x2000u

/* where a match came from (top three bits of a token) */
#define GLZ_SEL_P1 0u
#define GLZ_SEL_P2 1u
#define GLZ_SEL_TABLE 2u
#define GLZ_SEL_RUN 3u
#define GLZ_SEL_DEEP 4u
#define GLZ_TOK_LEN_MASK 0x1fffffffu

/* Pointers that arrive inside the job record are generic to the compiler; the
 * kernel retypes them as global memory so that loads through them are plain
 * global loads and values read at a wave-uniform address stay wave-uniform. */
#define GLZ_GLOBAL SPICE_GLOBAL
#define GLZ_EIGHT_PER_CU SPICE_WAVES_PER_EU(8) /* 64 VGPRs */
/* Hides where a lane index came from: what is derived from it is worked out again in
 * every window (an instruction or two) instead of being hoisted out of the window loop
 * into registers that the 64-register budget then spills to scratch memory. */
#define GLZ_RECOMPUTE(v) SPICE_OPAQUE_VGPR(v)
typedef uint32_t glz_u32x4 __attribute__((vector_size(16)));
typedef glz_u32x4 glz_u32x4_u __attribute__((aligned(4))); /* pixel-aligned only */

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
__device__ __forceinline__ uint32_t glz_load_px(const GLZ_GLOBAL uint8_t *base, int idx)
{
    if (MODE == GLZ_MODE_RGB32) {
        return ((const GLZ_GLOBAL uint32_t *)base)[idx];
    } else if (MODE == GLZ_MODE_RGB24) {
        const GLZ_GLOBAL uint8_t *p = base + (size_t)idx * 3;
        return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
    } else if (MODE == GLZ_MODE_RGB16) {
        return ((const GLZ_GLOBAL uint16_t *)base)[idx];
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

/* cap = GLZ_NO_CAP, written as that constant: the caller has checked that the whole token
 * fits, the test folds away and the store is plain (a test per byte is a compare, two
 * exec-mask instructions and a branch, and a match token has up to nine bytes) */
#define GLZ_NO_CAP 0xffffffffu

__device__ __forceinline__ void glz_put(GLZ_GLOBAL uint8_t *buf, uint32_t cap, uint32_t pos, uint32_t b)
{
    if (cap == GLZ_NO_CAP || pos < cap) {
        buf[pos] = (uint8_t)b;
    }
}

template <int MODE>
__device__ __forceinline__ void glz_put_pixel(GLZ_GLOBAL uint8_t *buf, uint32_t cap, uint32_t pos, uint32_t raw)
{
    if (MODE == GLZ_MODE_RGB32 || MODE == GLZ_MODE_RGB24) {
        glz_put(buf, cap, pos, raw & 0xff);
        glz_put(buf, cap, pos + 1, (raw >> 8) & 0xff);
        glz_put(buf, cap, pos + 2, (raw >> 16) & 0xff);
    } else if (MODE == GLZ_MODE_RGB16) {
        glz_put(buf, cap, pos, (raw >> 8) & 0xff);
        glz_put(buf, cap, pos + 1, raw & 0xff);
    } else {
        glz_put(buf, cap, pos, raw & 0xff);
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

/* glz_encode_match_tmpl.c:31-151: bytes of one match; len is already biased */
__device__ __forceinline__ uint32_t glz_match_bytes(uint32_t image_dist, uint32_t pix_dist, uint32_t len)
{
    const uint32_t nlen = len < 7 ? 1u : 2u + (len - 7) / 255u;
    return nlen + (uint32_t)glz_ref_bytes(image_dist, pix_dist) - 1;
}

/* glz_encode_match_tmpl.c:31-151, written by one lane.  The 0xff bytes of a
 * long length are left to the caller when there are more than max_ff of them. */
__device__ __forceinline__ void glz_write_match(GLZ_GLOBAL uint8_t *buf, uint32_t cap, uint32_t pos,
                                                uint32_t image_dist, uint32_t pix_dist,
                                                uint32_t len, uint32_t max_ff)
{
    const uint32_t far_px = pix_dist >= 4096u;
    const uint32_t low = (far_px ? 16u : 0u) + (pix_dist & 0x0f);
    uint32_t p = pos;

    if (len < 7) {
        glz_put(buf, cap, p++, (len << 5) + low);
    } else {
        const uint32_t rem = len - 7, nff = rem / 255u, last = rem - nff * 255u;
        glz_put(buf, cap, p++, (7u << 5) + low);
        if (nff <= max_ff) {
            for (uint32_t t = 0; t < nff; t++) {
                glz_put(buf, cap, p + t, 255u);
            }
        }
        p += nff;
        glz_put(buf, cap, p++, last);
    }
    glz_put(buf, cap, p++, (pix_dist >> 4) & 255u);
    if (!far_px) {
        if (image_dist < (1u << 6)) {
            glz_put(buf, cap, p++, image_dist & 0x3f);
        } else if (image_dist < (1u << 14)) {
            glz_put(buf, cap, p++, (1u << 6) + (image_dist & 0x3f));
            glz_put(buf, cap, p++, (image_dist >> 6) & 255u);
        } else if (image_dist < (1u << 22)) {
            glz_put(buf, cap, p++, (1u << 7) + (image_dist & 0x3f));
            glz_put(buf, cap, p++, (image_dist >> 6) & 255u);
            glz_put(buf, cap, p++, (image_dist >> 14) & 255u);
        } else {
            glz_put(buf, cap, p++, (1u << 7) + (1u << 6) + (image_dist & 0x3f));
            glz_put(buf, cap, p++, (image_dist >> 6) & 255u);
            glz_put(buf, cap, p++, (image_dist >> 14) & 255u);
            glz_put(buf, cap, p++, (image_dist >> 22) & 255u);
        }
    } else {
        const uint32_t very_far = pix_dist < 131072u ? 0u : 32u;
        const uint32_t hi = (pix_dist >> 12) & 31u;
        if (image_dist == 0) {
            glz_put(buf, cap, p++, very_far + hi);
        } else if (image_dist < (1u << 8)) {
            glz_put(buf, cap, p++, very_far + (1u << 6) + hi);
            glz_put(buf, cap, p++, image_dist & 255u);
        } else if (image_dist < (1u << 16)) {
            glz_put(buf, cap, p++, very_far + (1u << 7) + hi);
            glz_put(buf, cap, p++, image_dist & 255u);
            glz_put(buf, cap, p++, (image_dist >> 8) & 255u);
        } else {
            glz_put(buf, cap, p++, very_far + (1u << 7) + (1u << 6) + hi);
            glz_put(buf, cap, p++, image_dist & 255u);
            glz_put(buf, cap, p++, (image_dist >> 8) & 255u);
            glz_put(buf, cap, p++, (image_dist >> 16) & 255u);
        }
        if (very_far) {
            glz_put(buf, cap, p++, (pix_dist >> 17) & 255u);
        }
    }
}

__device__ __forceinline__ uint32_t glz_readlane(uint32_t v, int lane)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}

/* a value every lane of the wavefront holds, moved to where the compiler knows it */
__device__ __forceinline__ uint32_t glz_uniform(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}

__device__ __forceinline__ uint64_t glz_uniform64(uint64_t v)
{
    return (uint64_t)glz_uniform((uint32_t)v) | ((uint64_t)glz_uniform((uint32_t)(v >> 32)) << 32);
}

#define GLZ_ROWS 4 /* rows of 256 (4-byte pixels) or 64 (others) per iteration of the long compares */

/* the bits of a raw 4-byte pixel that take part in comparisons */
template <int MODE>
__device__ __forceinline__ uint32_t glz_raw_mask()
{
    return MODE == GLZ_MODE_ALPHA ? 0xff000000u : 0x00ffffffu;
}

/* leading pixels (0..4) on which two groups of four raw pixels agree */
__device__ __forceinline__ int glz_agree4(const glz_u32x4 a, const glz_u32x4 b, uint32_t mask)
{
    const bool e0 = ((a[0] ^ b[0]) & mask) == 0, e1 = ((a[1] ^ b[1]) & mask) == 0;
    const bool e2 = ((a[2] ^ b[2]) & mask) == 0, e3 = ((a[3] ^ b[3]) & mask) == 0;
    return !e0 ? 0 : (!e1 ? 1 : (!e2 ? 2 : (!e3 ? 3 : 4)));
}

/* Number of leading positions k in [start, maxn) with src[sidx+k] == ref[ridx+k],
 * plus start.  One wavefront; every load of an iteration is in flight before
 * the first compare.  Lanes past the end re-read the last whole group; no
 * byte beyond position maxn - 1 of either side is touched. */
template <int MODE, bool BUDGETED = false>
__device__ __forceinline__ int glz_match_len(const GLZ_GLOBAL uint8_t *sbase, int sidx,
                                             const GLZ_GLOBAL uint8_t *rbase, int ridx, int start,
                                             int maxn, int lane, int max_rounds = 0, bool *gave_up = nullptr)
{
    /* BUDGETED: after max_rounds round trips without a difference the compare gives up;
     * *gave_up says so, and the positions before the returned one are known to agree */
    typedef GlzTraits<MODE> T;
    int n = start;
    int rounds = 0;
    bool out_of_rounds = false;
    if (MODE == GLZ_MODE_RGB32 || MODE == GLZ_MODE_ALPHA) {
        const uint32_t mask = glz_raw_mask<MODE>();
        const int lim = maxn - 3; /* a group of four never reaches past maxn - 1 */
        while (n < lim) {
            glz_u32x4 a[GLZ_ROWS], b[GLZ_ROWS];
            int k[GLZ_ROWS];
#pragma unroll
            for (int u = 0; u < GLZ_ROWS; u++) {
                k[u] = n + u * 256 + lane * 4;
                const int kk = k[u] < lim ? k[u] : lim - 1;
                a[u] = *(const GLZ_GLOBAL glz_u32x4_u *)(sbase + (size_t)(sidx + kk) * 4);
                b[u] = *(const GLZ_GLOBAL glz_u32x4_u *)(rbase + (size_t)(ridx + kk) * 4);
            }
            int sub[GLZ_ROWS];
            uint64_t stop[GLZ_ROWS];
#pragma unroll
            for (int u = 0; u < GLZ_ROWS; u++) {
                int valid = lim - k[u];
                valid = valid < 0 ? 0 : (valid > 4 ? 4 : valid);
                const int g = glz_agree4(a[u], b[u], mask);
                sub[u] = g < valid ? g : valid;
                stop[u] = __ballot(sub[u] < 4);
            }
            bool done = false;
#pragma unroll
            for (int u = 0; u < GLZ_ROWS; u++) {
                if (!done) {
                    if (stop[u]) {
                        const int f = __ffsll((unsigned long long)stop[u]) - 1;
                        n += f * 4 + (int)glz_readlane((uint32_t)sub[u], f);
                        done = true;
                    } else {
                        n += 256;
                    }
                }
            }
            if (done) {
                break;
            }
            if (BUDGETED && ++rounds >= max_rounds && n < lim) {
                out_of_rounds = true;
                break;
            }
        }
        if (n >= lim) {
            /* the last three pixels one by one */
            n = lim > start ? lim : start;
            const int k = n + lane;
            bool differ = true;
            if (k < maxn) {
                differ = ((glz_load_px<MODE>(sbase, sidx + k) ^ glz_load_px<MODE>(rbase, ridx + k)) & T::MASK) != 0;
            }
            n += __ffsll((unsigned long long)__ballot(differ)) - 1;
        }
    } else {
        while (n < maxn) {
            uint32_t a[GLZ_ROWS * 2], b[GLZ_ROWS * 2];
#pragma unroll
            for (int u = 0; u < GLZ_ROWS * 2; u++) {
                const int k = n + u * 64 + lane;
                const int kk = k < maxn ? k : maxn - 1;
                a[u] = glz_load_px<MODE>(sbase, sidx + kk);
                b[u] = glz_load_px<MODE>(rbase, ridx + kk);
            }
            uint64_t stop[GLZ_ROWS * 2];
#pragma unroll
            for (int u = 0; u < GLZ_ROWS * 2; u++) {
                const int k = n + u * 64 + lane;
                stop[u] = __ballot(k >= maxn || ((a[u] ^ b[u]) & T::MASK) != 0);
            }
            bool done = false;
#pragma unroll
            for (int u = 0; u < GLZ_ROWS * 2; u++) {
                if (!done) {
                    if (stop[u]) {
                        n += __ffsll((unsigned long long)stop[u]) - 1;
                        done = true;
                    } else {
                        n += 64;
                    }
                }
            }
            if (done) {
                break;
            }
            if (BUDGETED && ++rounds >= max_rounds && n < maxn) {
                out_of_rounds = true;
                break;
            }
        }
    }
    if (BUDGETED) {
        *gave_up = out_of_rounds;
    }
    return n < maxn ? n : maxn;
}

/* Number of leading positions k < maxn with src[sidx+k] == x (x as glz_load_px
 * returns it).  One wavefront. */
template <int MODE>
__device__ __forceinline__ int glz_run_len(const GLZ_GLOBAL uint8_t *sbase, int sidx, uint32_t x,
                                           int maxn, int lane)
{
    typedef GlzTraits<MODE> T;
    int n = 0;
    if (MODE == GLZ_MODE_RGB32 || MODE == GLZ_MODE_ALPHA) {
        const uint32_t mask = glz_raw_mask<MODE>();
        const uint32_t xr = MODE == GLZ_MODE_ALPHA ? (x & 0xffu) << 24 : x;
        glz_u32x4 xs;
        xs[0] = xs[1] = xs[2] = xs[3] = xr;
        const int lim = maxn - 3; /* a group of four never reaches past maxn - 1 */
        while (n < lim) {
            glz_u32x4 a[GLZ_ROWS];
            int k[GLZ_ROWS];
#pragma unroll
            for (int u = 0; u < GLZ_ROWS; u++) {
                k[u] = n + u * 256 + lane * 4;
                const int kk = k[u] < lim ? k[u] : lim - 1;
                a[u] = *(const GLZ_GLOBAL glz_u32x4_u *)(sbase + (size_t)(sidx + kk) * 4);
            }
            int sub[GLZ_ROWS];
            uint64_t stop[GLZ_ROWS];
#pragma unroll
            for (int u = 0; u < GLZ_ROWS; u++) {
                int valid = lim - k[u];
                valid = valid < 0 ? 0 : (valid > 4 ? 4 : valid);
                const int g = glz_agree4(a[u], xs, mask);
                sub[u] = g < valid ? g : valid;
                stop[u] = __ballot(sub[u] < 4);
            }
            bool done = false;
#pragma unroll
            for (int u = 0; u < GLZ_ROWS; u++) {
                if (!done) {
                    if (stop[u]) {
                        const int f = __ffsll((unsigned long long)stop[u]) - 1;
                        n += f * 4 + (int)glz_readlane((uint32_t)sub[u], f);
                        done = true;
                    } else {
                        n += 256;
                    }
                }
            }
            if (done) {
                break;
            }
        }
        if (n >= lim) {
            /* the last three pixels one by one */
            n = lim > 0 ? lim : 0;
            const int k = n + lane;
            bool differ = true;
            if (k < maxn) {
                differ = ((glz_load_px<MODE>(sbase, sidx + k) ^ x) & T::MASK) != 0;
            }
            n += __ffsll((unsigned long long)__ballot(differ)) - 1;
        }
    } else {
        while (n < maxn) {
            uint32_t a[GLZ_ROWS * 2];
#pragma unroll
            for (int u = 0; u < GLZ_ROWS * 2; u++) {
                const int k = n + u * 64 + lane;
                const int kk = k < maxn ? k : maxn - 1;
                a[u] = glz_load_px<MODE>(sbase, sidx + kk);
            }
            uint64_t stop[GLZ_ROWS * 2];
#pragma unroll
            for (int u = 0; u < GLZ_ROWS * 2; u++) {
                const int k = n + u * 64 + lane;
                stop[u] = __ballot(k >= maxn || ((a[u] ^ x) & T::MASK) != 0);
            }
            bool done = false;
#pragma unroll
            for (int u = 0; u < GLZ_ROWS * 2; u++) {
                if (!done) {
                    if (stop[u]) {
                        n += __ffsll((unsigned long long)stop[u]) - 1;
                        done = true;
                    } else {
                        n += 64;
                    }
                }
            }
            if (done) {
                break;
            }
        }
    }
    return n < maxn ? n : maxn;
}

/* hash of the three pixels at idx, loaded afresh */
template <int MODE>
__device__ __forceinline__ uint32_t glz_hash_at(const GLZ_GLOBAL uint8_t *base, int idx)
{
    typedef GlzTraits<MODE> T;
    return glz_hash3<MODE>(glz_load_px<MODE>(base, idx) & T::MASK,
                           glz_load_px<MODE>(base, idx + 1) & T::MASK,
                           glz_load_px<MODE>(base, idx + 2) & T::MASK);
}

/* LDS shared by the workgroup */
#ifdef GLZ_LONE
/* Barrier of the wavefronts that share a phase (four of them): every wavefront adds one to
 * the group's counter in LDS and waits until the counter has reached what it was before plus
 * four.  The counter only grows; `target` is what the calling lane waits for next.  The
 * wavefronts of a group must all make the same barrier calls. */
__device__ __forceinline__ void glz_group_barrier(uint32_t *ctr, uint32_t &target, int lane)
{
    spice_group_barrier(ctr, target, lane, 4);
}

#ifdef GLZ_LONE_STAMPS
#define LONE_STAMP() clock64()
#define LONE_ADD(k, v) do { if (lane == 0) glz_sh.st_x[k] += (v); } while (0)
#define LONE_CLASS(c) do { if (lane == 0) glz_sh.st_class = (c); } while (0)
#else
#define LONE_CLASS(c) do { } while (0)
#endif
/* One group of wavefronts waits for what another has done: the doers add to a counter in LDS
 * once their (global) stores are out, the waiters poll it. */
__device__ __forceinline__ void glz_signal(uint32_t *ctr, int lane)
{
    spice_signal(ctr, lane);
}
__device__ __forceinline__ void glz_wait_for(uint32_t *ctr, uint32_t target)
{
    spice_wait_for(ctr, target);
}

/* The kernel for a GPU with few sessions keeps three windows in LDS: one being scanned,
 * one being walked, one being written out (glz_lone.hip). */
#define GLZ_HEADS 4
#define GLZ_HEAD_ROUNDS 8              /* round trips a compare ahead of the walk may take */
struct GlzWin {                        /* what one window needs from its scan to its emit */
    uint32_t tile[GLZ_TILE];
    uint32_t hash[GLZ_WIN];
    uint32_t bitmap[GLZ_KEYS * 8];
    uint32_t infoA[GLZ_WIN];
    uint32_t infoB[GLZ_WIN];
    uint32_t g_idist[GLZ_WIN];
    uint32_t g_pdist[GLZ_WIN];
    uint32_t g_ref[GLZ_WIN];
    uint32_t g_seg[GLZ_WIN];
    uint32_t tok[GLZ_WIN];
    uint32_t tok_pdist[GLZ_WIN];
    uint64_t lit[4], ms[4], ins[4];
    uint64_t neq[4];
    uint64_t slow[4];
    uint32_t tile_m2;
    uint32_t pend;
    int next_ip;
    int tail0, tail1;                  /* where the walk's last match stored behind itself, window positions (-1: nowhere) */
    int start_ip;                      /* the window's first position, as the scanning wavefronts chose it */
    int jump;                          /* ... 1: where a long match of the window in front ends; 0: right behind that window */
    /* Long matches against the table's candidates, measured ahead of the walk (glz_window_heads):
     * a head is the first of consecutive positions whose candidates lie at one distance in one
     * segment -- one match seen from each of them -- and ends where that match ends. */
    uint64_t headmask[4];
    int head_pos[GLZ_HEADS];           /* window position, -1: none */
    uint32_t head_seg[GLZ_HEADS];      /* the candidates' segment */
    int head_delta[GLZ_HEADS];         /* candidate's pixel index minus the position's */
    int head_end[GLZ_HEADS];           /* first position of this segment that differs (or the limit) */
    int head_done[GLZ_HEADS];          /* 0: the compare gave up at head_end, the match may go on */
    uint32_t heads_ready;              /* bit h: head h's record is complete (it may be written while the window is walked) */
};
#define GLZ_SLOTS 3
struct GlzShared {
    GlzWin win[GLZ_SLOTS];
    GlzDevSeg seg_cache[GLZ_SEG_CACHE];
    uint32_t wave_sum[4];
    uint32_t op, copy, last_copy;      /* output cursor, see below */
    uint32_t arrive[2];                /* arrivals at the barriers of the scanning and the emitting wavefronts */
    uint32_t stores_made;              /* wavefronts that have made their share of a window's table stores, ever */
#ifdef GLZ_LONE_STAMPS
    long long st_fast, st_gen[5], st_ngen[5]; /* the walk: fast loop; general path by kind */
    long long st_x[4];                        /* ... quarter set-up, quarter end, the rest of a window */
    int st_class;
#endif
};
static_assert(sizeof(GlzShared) <= 64 * 1024, "one workgroup's static LDS");
#define GLZ_W win
#define GLZ_WIN_ARG GlzWin &win,
#define GLZ_WIN_PASS win,
/* the walk enters a window at position entry_i with the table stores of the match that brought
 * it there at positions entry_t0 / entry_t1 (-1: none) */
#define GLZ_ENTRY_ARGS , int entry_i, int entry_t0, int entry_t1
#define GLZ_ENTRY_I entry_i
#define GLZ_ENTRY_T0 entry_t0
#define GLZ_ENTRY_T1 entry_t1
/* a phase's own wavefronts meet at a barrier of their own (the others are busy elsewhere) */
#define GLZ_PHASE_BARRIER(which) glz_group_barrier(&glz_sh.arrive[which], grp_target, lane)
#define GLZ_GROUP_ARG , uint32_t &grp_target
/* stores: this call makes the window's table stores and nothing else; otherwise it writes the
 * window's tokens and makes no table store */
#define GLZ_EMIT_ARG , uint32_t &grp_target, bool stores, uint32_t later_h0, uint32_t later_h1
/* later_h0 / later_h1: slots that will be stored to right after this window's stores (the
 * positions behind the match that led out of it), GLZ_NO_HASH for none */
#define GLZ_DOMINATED_AT_ONCE (my_h == later_h0 || my_h == later_h1)
#else
struct GlzShared {
    uint32_t tile[GLZ_TILE];           /* raw pixels of positions ip-1 .. */
    uint32_t hash[GLZ_WIN];            /* 3-pixel hash of every window position */
    uint32_t bitmap[GLZ_KEYS * 8];     /* row = hash & (KEYS-1): window positions with that key */
    uint32_t infoA[GLZ_WIN];           /* scan results, GLZ_A_* / GLZ_B_* */
    uint32_t infoB[GLZ_WIN];
    uint32_t g_idist[GLZ_WIN];         /* the table entry's candidate */
    uint32_t g_pdist[GLZ_WIN];
    uint32_t g_ref[GLZ_WIN];
    uint32_t g_seg[GLZ_WIN];
    uint32_t tok[GLZ_WIN];             /* per match start: length | GLZ_SEL_* << 29 */
    uint32_t tok_pdist[GLZ_WIN];       /* pixel distance of a GLZ_SEL_DEEP match */
    GlzDevSeg seg_cache[GLZ_SEG_CACHE];
    uint64_t lit[4], ms[4], ins[4];    /* window positions: literal stops, match starts, table stores */
    uint64_t neq[4];                   /* ... pixel differs from its predecessor */
    uint64_t slow[4];                  /* ... stops whose outcome wavefront 0 left in tok[] */
    uint32_t wave_sum[4];
    uint32_t tile_m2;                  /* raw pixel of position ip-2 */
    uint32_t pend;                     /* table stores the last match left for positions ip-2 (bit 0) and ip-1 (bit 1) */
    int next_ip;
    /* output cursor.  Literal runs are written lazily: a count byte goes out
     * together with the first literal of every 32 (glz_encoder.c:89-108 writes
     * it ahead and patches or drops it; the bytes are the same) */
    uint32_t op;                       /* next byte */
    uint32_t copy;                     /* literals in the open count byte's run, 0 = none open */
    uint32_t last_copy;                /* where that count byte is */
#ifdef GLZ_STAMPS
    long long st_fast, st_gen, st_ngen;
    long long st_scan[8];
    long long st_emit[8];
#endif
};

static_assert(sizeof(GlzShared) <= 20 * 1024, "eight workgroups per CU share 160 KB of LDS");
#define GLZ_W sh
#define GLZ_WIN_ARG
#define GLZ_WIN_PASS
#define GLZ_ENTRY_ARGS
#define GLZ_ENTRY_I 0
#define GLZ_ENTRY_T0 -1
#define GLZ_ENTRY_T1 -1
#define GLZ_PHASE_BARRIER(which) __syncthreads()
#define GLZ_GROUP_ARG
#define GLZ_EMIT_ARG
#define GLZ_DOMINATED_AT_ONCE false
#define LONE_CLASS(c) do { } while (0)
#endif
__shared__ GlzShared glz_sh;
#ifdef GLZ_STAMPS
__device__ unsigned long long glz_stamp_acc[24];
/* when each workgroup of the last launch started and ended (100 MHz clock) */
__device__ unsigned long long glz_wg_times[8192 * 2];
extern "C" void spice_gpu_debug_wg_times(unsigned long long *out, int n)
{
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(glz_wg_times), sizeof(unsigned long long) * 2 * (size_t)n);
}
extern "C" void spice_gpu_debug_stamps(unsigned long long *out, int reset)
{
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(glz_stamp_acc), sizeof(unsigned long long) * 24);
    if (reset) {
        unsigned long long z[24] = {0};
        hipMemcpyToSymbol(HIP_SYMBOL(glz_stamp_acc), z, sizeof(z));
    }
}
#define STAMP() clock64()
#define SCAN_STAMP(i) do { const long long n_ = STAMP(); if (tid == 0) sh.st_scan[i] += n_ - ss_; ss_ = n_; } while (0)
#define EMIT_STAMP(i) do { const long long n_ = STAMP(); if (tid == 0) sh.st_emit[i] += n_ - es_; es_ = n_; } while (0)
#else
#define SCAN_STAMP(i) do { } while (0)
#define EMIT_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ const GLZ_GLOBAL uint8_t *glz_arena(const GlzDevJob &job)
{
    return (const GLZ_GLOBAL uint8_t *)(uintptr_t)job.arena;
}

/* Where a segment's pixels are.  A single-segment image of a device-resident
 * batch is parsed straight from the caller's buffer while its copy into the
 * window arena (needed by later images only) runs beside the parse. */
__device__ __forceinline__ const GLZ_GLOBAL uint8_t *glz_seg_base(const GlzDevJob &job, uint32_t seg_idx,
                                                                  const GlzDevSeg &s)
{
    const GLZ_GLOBAL uint8_t *in_arena = glz_arena(job) + (size_t)s.arena_off16 * 16;
    const GLZ_GLOBAL uint8_t *own = (const GLZ_GLOBAL uint8_t *)(uintptr_t)job.own_src;
    return (job.own_src != nullptr && seg_idx == job.own_seg) ? own : in_arena;
}

__device__ __forceinline__ GLZ_GLOBAL uint8_t *glz_out(const GlzDevJob &job)
{
    return (GLZ_GLOBAL uint8_t *)(uintptr_t)job.out;
}

/* a table entry is {segment, pixel}: one aligned 8-byte word, segment in the low half */
__device__ __forceinline__ uint2 glz_htab_load(const GlzDevJob &job, uint32_t h)
{
    const uint64_t e = ((const GLZ_GLOBAL uint64_t *)(uintptr_t)job.htab)[h];
    return make_uint2((uint32_t)e, (uint32_t)(e >> 32));
}

__device__ __forceinline__ void glz_htab_store(const GlzDevJob &job, uint32_t h, uint32_t seg, uint32_t pix)
{
    ((GLZ_GLOBAL uint64_t *)(uintptr_t)job.htab)[h] = (uint64_t)seg | ((uint64_t)pix << 32);
}

__device__ __forceinline__ GlzDevSeg glz_load_seg(const GlzDevJob &job, uint32_t idx)
{
    const GLZ_GLOBAL glz_u32x4 *p = (const GLZ_GLOBAL glz_u32x4 *)(uintptr_t)(job.segs + idx);
    const glz_u32x4 a = p[0], b = p[1];
    GlzDevSeg s;
    s.pixels_so_far = (uint64_t)a[0] | ((uint64_t)a[1] << 32);
    s.image_id = (uint64_t)a[2] | ((uint64_t)a[3] << 32);
    s.arena_off16 = b[0];
    s.nunits = b[1];
    s.img_first_delta = b[2];
    s.flags = b[3];
    return s;
}

__device__ __forceinline__ void glz_store_seg(const GlzDevJob &job, uint32_t idx, const GlzDevSeg &s)
{
    GLZ_GLOBAL glz_u32x4 *p = (GLZ_GLOBAL glz_u32x4 *)(uintptr_t)(job.segs + idx);
    glz_u32x4 a, b;
    a[0] = (uint32_t)s.pixels_so_far;
    a[1] = (uint32_t)(s.pixels_so_far >> 32);
    a[2] = (uint32_t)s.image_id;
    a[3] = (uint32_t)(s.image_id >> 32);
    b[0] = s.arena_off16;
    b[1] = s.nunits;
    b[2] = s.img_first_delta;
    b[3] = s.flags;
    p[0] = a;
    p[1] = b;
}


__device__ __forceinline__ GlzDevSeg glz_seg(const GlzDevJob &job, const GlzShared &sh, uint32_t idx)
{
    if (idx < GLZ_SEG_CACHE) {
        return sh.seg_cache[idx];
    }
    return glz_load_seg(job, idx);
}

/* A segment record every lane asked for with the same index: held in scalar
 * registers from here on (as vector registers its eight dwords, live across the
 * whole window loop, are the first thing the register allocator spills) */
__device__ __forceinline__ GlzDevSeg glz_seg_uniform(const GlzDevSeg &v)
{
    GlzDevSeg s;
    s.pixels_so_far = glz_uniform64(v.pixels_so_far);
    s.image_id = glz_uniform64(v.image_id);
    s.arena_off16 = glz_uniform(v.arena_off16);
    s.nunits = glz_uniform(v.nunits);
    s.img_first_delta = glz_uniform(v.img_first_delta);
    s.flags = glz_uniform(v.flags);
    return s;
}

__device__ __forceinline__ uint32_t glz_pick4(uint32_t r0, uint32_t r1, uint32_t r2, uint32_t r3, int w)
{
    return w == 0 ? r0 : (w == 1 ? r1 : (w == 2 ? r2 : r3));
}

__device__ __forceinline__ uint64_t glz_pick4(uint64_t r0, uint64_t r1, uint64_t r2, uint64_t r3, int w)
{
    return w == 0 ? r0 : (w == 1 ? r1 : (w == 2 ? r2 : r3));
}

/* first set bit at or above position `from` of a 256-bit mask, GLZ_WIN if none */
__device__ __forceinline__ int glz_ffs_from(uint64_t m0, uint64_t m1, uint64_t m2, uint64_t m3, int from)
{
    if (from >= GLZ_WIN) {
        return GLZ_WIN;
    }
    const int w = from >> 6, s = from & 63;
    const uint64_t cur = (glz_pick4(m0, m1, m2, m3, w) >> s) << s;
    if (cur) {
        return w * 64 + __ffsll((unsigned long long)cur) - 1;
    }
    if (w < 1 && m1) {
        return 64 + __ffsll((unsigned long long)m1) - 1;
    }
    if (w < 2 && m2) {
        return 128 + __ffsll((unsigned long long)m2) - 1;
    }
    if (w < 3 && m3) {
        return 192 + __ffsll((unsigned long long)m3) - 1;
    }
    return GLZ_WIN;
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

/* what do_match accepts for a candidate inside the window: image distance 0,
 * pixel distance < 4096, so three reference bytes (glz_encode_tmpl.c:243-261) */
template <int MODE>
__device__ __forceinline__ int glz_accept_near(int len)
{
    typedef GlzTraits<MODE> T;
    const int need = 3 / T::ENC_DIV + 1;
    if (len < T::MIN_REF) {
        return 0;
    }
    return (len > T::MAX_REF || len >= need) ? len : 0;
}

/* Literal pixels [first, first + n) of a segment, n <= 64, appended to the
 * open literal run straight from the window arena (segment heads and tails,
 * glz_encode_tmpl.c:466-487,511-541).  closing: the run ends after them.
 * Wavefront 0 only. */
template <int MODE>
__device__ __forceinline__ void glz_run_from_arena(const GlzDevJob &job, const GLZ_GLOBAL uint8_t *base,
                                                   int first, int n, bool closing, int lane)
{
    typedef GlzTraits<MODE> T;
    GlzShared &sh = glz_sh;
    const uint32_t op = sh.op, c = sh.copy, lc = sh.last_copy;
    const bool active = lane < n;
    const uint32_t r = c + (uint32_t)lane;
    const bool cs = active && (r & 31) == 0;               /* starts a run of 32 */
    const uint64_t csmask = __ballot(cs);
    const uint32_t ctrl_before = (uint32_t)__popcll(csmask & ((1ull << lane) - 1));
    const uint32_t total = c + (uint32_t)n;
    uint32_t off = op + (uint32_t)lane * T::LITB + ctrl_before;

    if (active) {
        if (cs) {
            const uint32_t left = total - r;
            glz_put(glz_out(job), job.out_cap, off, closing ? (left < 32 ? left : 32) - 1 : GLZ_MAX_COPY - 1);
            off++;
        }
        glz_put_pixel<MODE>(glz_out(job), job.out_cap, off, glz_load_px<MODE>(base, first + lane));
    }
    if (lane == 0) {
        if (c > 0 && closing) {
            glz_put(glz_out(job), job.out_cap, lc, (total < 32 ? total : 32) - 1);
        }
        if (csmask) {
            const int kk = 63 - __clzll((long long)csmask);
            sh.last_copy = op + (uint32_t)kk * T::LITB + (uint32_t)__popcll(csmask & ((1ull << kk) - 1));
        }
        sh.op = op + (uint32_t)n * T::LITB + (uint32_t)__popcll(csmask);
        sh.copy = closing ? 0 : (total & 31);
    }
}

/* ---- scan: one window position per lane.  Leaves infoA/infoB and the table
 * candidate in LDS; returns the lane's hash (for the later table store). */
template <int MODE>
__device__ __forceinline__ void glz_window_scan(GLZ_WIN_ARG const GlzDevJob &job, const GlzDevSeg &cs,
                                                uint32_t seg_idx, const GLZ_GLOBAL uint8_t *base, int ip,
                                                int npix, int tid, uint32_t pend, uint32_t th0, uint32_t th1,
                                                uint32_t &my_h, bool &my_hv GLZ_GROUP_ARG)
{
    typedef GlzTraits<MODE> T;
    GlzShared &sh = glz_sh;
#ifdef GLZ_STAMPS
    long long ss_ = STAMP();
#endif
    const int ip_bound = npix - 2, ip_limit = npix - 6;
    GLZ_RECOMPUTE(tid);
    const int p = tid, q = ip + p;
    const bool act = q < ip_limit;
    const bool hv = q + 2 < npix;
    const uint32_t cm1 = GLZ_W.tile[p] & T::MASK, c0 = GLZ_W.tile[p + 1] & T::MASK;
    const uint32_t c1 = GLZ_W.tile[p + 2] & T::MASK, c2 = GLZ_W.tile[p + 3] & T::MASK;
    const uint32_t h = glz_hash3<MODE>(c0, c1, c2);
    const bool rle = act && q > 0 && cm1 == c0 && c0 == c1 && c1 == c2;
    const bool look = act && !rle;
    const uint32_t key = h & (GLZ_KEYS - 1);

    GLZ_W.hash[p] = h;
    if (hv) {
        atomicOr(&GLZ_W.bitmap[GLZ_BM(key, (uint32_t)p >> 5)], 1u << (p & 31));
    }
    /* the table probe travels while the window candidates are worked out */
    uint2 he = make_uint2(0, 0);
    if (look) {
        he = glz_htab_load(job, h);
    }
    my_h = h;
    my_hv = hv;
    {
        const uint64_t neq = __ballot(c0 != cm1);
        if ((tid & 63) == 0) {
            GLZ_W.neq[tid >> 6] = neq;
            GLZ_W.lit[tid >> 6] = 0;
            GLZ_W.ms[tid >> 6] = 0;
            GLZ_W.ins[tid >> 6] = 0;
            GLZ_W.slow[tid >> 6] = 0;
        }
    }
#ifdef GLZ_LONE
    const int lane = tid & 63;
#endif
    GLZ_PHASE_BARRIER(0);
    SCAN_STAMP(0);

    uint32_t A = (act ? GLZ_A_ACT : 0u) | (rle ? GLZ_A_RLE : 0u);
    uint32_t B = 0;
    const int maxn = ip_bound - q;

    /* (1) the two nearest earlier window positions with this hash.  (Reading the
     * key's whole row into registers first and walking it there was measured and
     * is slower: eight unrolled bit loops diverge worse than this one, and the row
     * costs eight registers.) */
    int n1 = -1, n2 = -1;
    if (look) {
        for (int v = p >> 5; v >= 0 && n2 < 0; v--) {
            uint32_t bits = GLZ_W.bitmap[GLZ_BM(key, (uint32_t)v)];
            if (v == (p >> 5)) {
                bits &= (1u << (p & 31)) - 1;
            }
            while (bits && n2 < 0) {
                const int b = 31 - __clz((int)bits);
                bits &= ~(1u << b);
                const int j = v * 32 + b;
                if (GLZ_W.hash[j] == h) {
                    if (n1 < 0) {
                        n1 = j;
                    } else {
                        n2 = j;
                    }
                }
            }
        }
    }

    SCAN_STAMP(1);
    /* (2) the table's entry: validation and do_match's distance rules
     * (glz_encode_tmpl.c:147-229); its pixels are requested here and compared
     * in (4) */
    /* everything but gok is read only where gok says it was set: left unset otherwise, so
     * that no path spends instructions on zeroing a dozen registers for the others */
    bool gok = false, capped;
    int maxk;
    uint32_t gidist, gpdist;
    uint32_t r[GLZ_K];
    SPICE_LEFT_UNSET(capped = false; maxk = 0; gidist = gpdist = 0; for (int k = 0; k < GLZ_K; k++) { r[k] = 0; });
    {
        /* Straight-line up to the pixel requests: every test is a select on values all
         * lanes work out (a lane that does not look at the table works on the zero entry
         * it holds and is masked out by `look` at the end), so that the exec mask is
         * touched once, round the requests themselves. */
        /* the stores the previous window's last match left for positions ip-2 and ip-1
         * are the newest entries of their slots, whatever the probe saw */
        const bool o1 = (pend & 2u) && h == th1, o0 = (pend & 1u) && h == th0;
        he.x = (o1 || o0) ? seg_idx : he.x;
        he.y = o1 ? (uint32_t)(ip - 1) : (o0 ? (uint32_t)(ip - 2) : he.y);
        const GlzDevSeg rs = glz_seg(job, sh, he.x);
        const bool valid =
            he.x == seg_idx ||
            ((rs.flags & GLZ_SEG_USED) && (rs.flags & GLZ_SEG_ALIVE) &&
             (rs.flags & GLZ_SEG_TYPE_MASK) == (cs.flags & GLZ_SEG_TYPE_MASK) &&
             rs.pixels_so_far <= cs.pixels_so_far && job.head_psf <= rs.pixels_so_far);
        const uint64_t idist = cs.image_id - rs.image_id;
        const uint64_t pdist = idist == 0 ? ((uint64_t)q + cs.pixels_so_far) - ((uint64_t)he.y + rs.pixels_so_far)
                                          : (uint64_t)he.y + rs.img_first_delta;
        const int ref = (int)he.y;
        const int rmax = (int)rs.nunits - ref;
        const int mk = rmax < maxn ? rmax : maxn;
        gok = look && valid && (int64_t)he.y <= (int64_t)rs.nunits - T::MIN_REF && pdist != 0 &&
              pdist < 33554432ull && idist <= 16777215ull;
        capped = mk > GLZ_K;
        maxk = capped ? GLZ_K : mk;
        gidist = (uint32_t)idist;
        gpdist = (uint32_t)pdist;
        if (gok) {
            const GLZ_GLOBAL uint8_t *rbase = glz_seg_base(job, he.x, rs);
            if (MODE == GLZ_MODE_RGB32 && GLZ_K == 8 && maxk == GLZ_K) {
                /* eight whole pixels lie ahead: two 16-byte requests instead of eight
                 * of four bytes -- every request of the wavefront, wide or narrow,
                 * is a pass through the CU's one address unit over up to 64 lines */
                const GLZ_GLOBAL glz_u32x4_u *w16 = (const GLZ_GLOBAL glz_u32x4_u *)(rbase + (size_t)ref * 4);
                const glz_u32x4 lo = w16[0], hi = w16[1];
                r[0] = lo[0];
                r[1] = lo[1];
                r[2] = lo[2];
                r[3] = lo[3];
                r[GLZ_K - 4] = hi[0];
                r[GLZ_K - 3] = hi[1];
                r[GLZ_K - 2] = hi[2];
                r[GLZ_K - 1] = hi[3];
            } else {
#pragma unroll
                for (int k = 0; k < GLZ_K; k++) {
                    r[k] = glz_load_px<MODE>(rbase, ref + (k < maxk ? k : 0));
                }
            }
        }
    }
    SCAN_STAMP(2);
    /* (3) the window candidates against the LDS tile */
    uint32_t mine[GLZ_K];
#pragma unroll
    for (int k = 0; k < GLZ_K; k++) {
        mine[k] = GLZ_W.tile[p + 1 + k] & T::MASK;
    }
    {
        /* both candidates' pixels are asked for in one go, whether a lane has them or not
         * (a lane without reads its own position), and what a lane lacks is masked out at
         * the end: one LDS round trip and no divergent blocks */
        const int q1 = n1 >= 0 ? n1 : p, q2 = n2 >= 0 ? n2 : p;
        int len1 = 0, len2 = 0;
        bool run1 = true, run2 = true;
#pragma unroll
        for (int k = 0; k < GLZ_K; k++) {
            const uint32_t t1 = GLZ_W.tile[q1 + 1 + k] & T::MASK, t2 = GLZ_W.tile[q2 + 1 + k] & T::MASK;
            run1 = run1 && k < maxn && t1 == mine[k];
            run2 = run2 && k < maxn && t2 == mine[k];
            len1 += run1 ? 1 : 0;
            len2 += run2 ? 1 : 0;
        }
        const bool maxed1 = len1 == GLZ_K && GLZ_K < maxn, maxed2 = len2 == GLZ_K && GLZ_K < maxn;
        const uint32_t a1 = GLZ_A_P1V | ((n1 >> 6) == (p >> 6) ? GLZ_A_P1SAME : 0u) | ((uint32_t)n1 << GLZ_A_P1_SHIFT) |
                            ((uint32_t)glz_accept_near<MODE>(len1) << GLZ_A_LEN1_SHIFT) | (maxed1 ? GLZ_A_MAX1 : 0u);
        /* a third position with this hash is not looked for: take it that one exists */
        const uint32_t a2 = GLZ_A_P2V | GLZ_A_MORE | ((n2 >> 6) == (p >> 6) ? GLZ_A_P2SAME : 0u);
        const uint32_t b2 = (uint32_t)n2 | ((uint32_t)glz_accept_near<MODE>(len2) << GLZ_B_LEN2_SHIFT) |
                            (maxed2 ? GLZ_B_MAX2 : 0u);
        A |= (n1 >= 0 ? a1 : 0u) | (n2 >= 0 ? a2 : 0u);
        B = n2 >= 0 ? b2 : B;
    }

    SCAN_STAMP(3);
    /* (4) the table's candidate */
    if (gok) {
        int len = 0;
        bool run = true;
#pragma unroll
        for (int k = 0; k < GLZ_K; k++) {
            run = run && k < maxk && (r[k] & T::MASK) == mine[k];
            len += run ? 1 : 0;
        }
        if (len >= T::MIN_REF && len <= T::MAX_REF) {
            /* glz_encode_tmpl.c:243-261 */
            const int need = glz_ref_bytes(gidist, gpdist) / T::ENC_DIV + 1;
            if (len < need) {
                len = 0;
            }
        }
        if (len >= T::MIN_REF) {
            A |= ((uint32_t)len << GLZ_A_GLEN_SHIFT) | (capped && len == GLZ_K ? GLZ_A_GMAX : 0u);
            GLZ_W.g_ref[p] = he.y;
            GLZ_W.g_seg[p] = he.x;
            GLZ_W.g_idist[p] = gidist;
            GLZ_W.g_pdist[p] = gpdist;
        }
    }
    SCAN_STAMP(4);
    if (rle) {
        /* glz_encode_tmpl.c:310-335: the run ends at the first pixel that
         * differs from its predecessor; B = its length, 0 when it leaves the
         * window (glz_resolve_general measures those) */
        const int from = p + 3, fw = from >> 6;
        const uint64_t keep = ~0ull << (from & 63);
        const uint64_t q0 = GLZ_W.neq[0], q1 = GLZ_W.neq[1], q2 = GLZ_W.neq[2], q3 = GLZ_W.neq[3];
        const uint64_t m0 = fw == 0 ? q0 & keep : 0ull;
        const uint64_t m1 = fw == 1 ? q1 & keep : (fw < 1 ? q1 : 0ull);
        const uint64_t m2 = fw == 2 ? q2 & keep : (fw < 2 ? q2 : 0ull);
        const uint64_t m3 = fw == 3 ? q3 & keep : (fw < 3 ? q3 : 0ull);
        int j = GLZ_WIN;
        if (m0) {
            j = __ffsll((unsigned long long)m0) - 1;
        } else if (m1) {
            j = 64 + __ffsll((unsigned long long)m1) - 1;
        } else if (m2) {
            j = 128 + __ffsll((unsigned long long)m2) - 1;
        } else if (m3) {
            j = 192 + __ffsll((unsigned long long)m3) - 1;
        }
        if (fw <= 3 && j < GLZ_WIN) {
            int end = ip + j;
            end = end < ip_bound ? end : ip_bound;
            B = (uint32_t)(end - q);
        }
    }
    GLZ_W.infoA[p] = A;
    GLZ_W.infoB[p] = B;
    SCAN_STAMP(5);
}

/* stores of one match's tail (window positions t0, t1; -1 = none) that fall into quarter q */
__device__ __forceinline__ uint64_t glz_tail_bits(int t0, int t1, int q)
{
    uint64_t m = 0;
    if (t0 >= 0 && (t0 >> 6) == q) {
        m |= 1ull << (t0 & 63);
    }
    if (t1 >= 0 && (t1 >> 6) == q) {
        m |= 1ull << (t1 & 63);
    }
    return m;
}

/* What wavefront 0 keeps per position of the quarter it is walking (one lane
 * each): the candidates whose fate is not known yet -- earlier positions of the
 * same quarter -- and what the stop falls back to when neither was stored.
 * Candidates in finished quarters are settled when the quarter is entered. */
#define GLZ_W_D2_SHIFT 6      /* 6 bits each: quarter-local index of the candidate, own index = none */
#define GLZ_W_L1_SHIFT 12     /* 5 bits each: accepted length, or a GLZ_W_LEN_* code */
#define GLZ_W_L2_SHIFT 17
#define GLZ_W_LF_SHIFT 22
#define GLZ_W_SLOW 0x80000000u /* no stop here, or a run that leaves the window */
#define GLZ_W_RUN 0x40000000u  /* the run test succeeds; the low bits hold the run's length */
#define GLZ_W_LEN_MAXED 30     /* the compare has to go on past GLZ_K */
#define GLZ_W_LEN_DEEP 31      /* an older store to the slot may exist */

template <int MODE>
__device__ __forceinline__ uint32_t glz_quarter_words(GLZ_WIN_ARG int Q, int lane, uint32_t &o_a, uint32_t &o_b)
{
    GlzShared &sh = glz_sh;
    const uint32_t *ins32 = (const uint32_t *)GLZ_W.ins;
    const int p = Q * 64 + lane;
    const uint32_t a = GLZ_W.infoA[p], b = GLZ_W.infoB[p];
    o_a = a; /* the walk asks for them by lane index instead of going to LDS again */
    o_b = b;
    if (!(a & GLZ_A_ACT)) {
        return GLZ_W_SLOW;
    }
    if (a & GLZ_A_RLE) {
        return b ? GLZ_W_RUN | b : GLZ_W_SLOW; /* a run that leaves the window: glz_resolve_general */
    }
    uint32_t d1 = (uint32_t)lane, d2 = (uint32_t)lane, l1 = 0, l2 = 0;
    uint32_t lf = (a & GLZ_A_GMAX) ? GLZ_W_LEN_MAXED : (a >> GLZ_A_GLEN_SHIFT) & 31u;
    if (a & GLZ_A_P1V) {
        const uint32_t n1 = (a >> GLZ_A_P1_SHIFT) & 255u, n2 = b & 255u;
        const uint32_t len1 = (a & GLZ_A_MAX1) ? GLZ_W_LEN_MAXED : (a >> GLZ_A_LEN1_SHIFT) & 31u;
        const uint32_t len2 = (b & GLZ_B_MAX2) ? GLZ_W_LEN_MAXED : (b >> GLZ_B_LEN2_SHIFT) & 31u;
        const bool p2v = a & GLZ_A_P2V, more = a & GLZ_A_MORE;
        if (a & GLZ_A_P1SAME) {
            d1 = n1 & 63u;
            l1 = len1;
            if (p2v) {
                if (a & GLZ_A_P2SAME) {
                    d2 = n2 & 63u;
                    l2 = len2;
                    lf = more ? GLZ_W_LEN_DEEP : lf;
                } else if ((ins32[n2 >> 5] >> (n2 & 31u)) & 1u) {
                    lf = len2;
                } else if (more) {
                    lf = GLZ_W_LEN_DEEP;
                }
            }
        } else if ((ins32[n1 >> 5] >> (n1 & 31u)) & 1u) {
            lf = len1;
        } else if (p2v) {
            if ((ins32[n2 >> 5] >> (n2 & 31u)) & 1u) {
                lf = len2;
            } else if (more) {
                lf = GLZ_W_LEN_DEEP;
            }
        }
    }
    return d1 | (d2 << GLZ_W_D2_SHIFT) | (l1 << GLZ_W_L1_SHIFT) | (l2 << GLZ_W_L2_SHIFT) | (lf << GLZ_W_LF_SHIFT);
}

/* The walk over the ordinary stops of a quarter, from position l on: picks the
 * length (the candidate at d1 if that position was stored, else the one at d2
 * if stored, else the fall-back), marks the stop, its store and, for a match,
 * the stores of its last two positions (glz_encode_tmpl.c:381,424-444: for
 * every pixel type a match of 2 stores one position, a longer one two), and
 * moves on.  Returns at the quarter's end (l >= 64) or at a stop that needs
 * glz_resolve_general.  A wavefront runs this at one instruction per ~5
 * cycles, so the loop is written out instruction by instruction. */
#include <spice_target_glz_walk.h> /* glz_fast_stops: written for the target */

/* The stops that need more than the quarter word: runs, compares that go on
 * past GLZ_K, slots whose two nearest predecessors were both skipped.  Returns
 * the length; *o_run = the run test succeeded (the stop stores nothing).
 * ins = stores of the current quarter so far, vfl = those of finished quarters
 * (lane l, bit q = position 64 q + l). */
template <int MODE>
__device__ __forceinline__ int glz_resolve_general(GLZ_WIN_ARG const GlzDevJob &job, const GLZ_GLOBAL uint8_t *base,
                                                   int ip, int ip_bound, int i, uint32_t a, uint32_t b,
                                                   uint64_t ins, uint32_t vfl, uint64_t q0, uint64_t q1,
                                                   uint64_t q2, uint64_t q3, int lane, bool *o_run)
{
    GlzShared &sh = glz_sh;
    const int l = i & 63, Q = i >> 6;
    const int anchor = ip + i;
    int len = 0;
    uint32_t sel = GLZ_SEL_RUN;

    *o_run = a & GLZ_A_RLE;
    LONE_CLASS(4);
    if (a & GLZ_A_RLE) {
        LONE_CLASS(0);
        /* glz_encode_tmpl.c:310-335: the run ends at the first pixel that
         * differs from its predecessor */
        const int j = glz_ffs_from(q0, q1, q2, q3, i + 3);
        int end;
        if (j < GLZ_WIN) {
            end = ip + j;
            end = end < ip_bound ? end : ip_bound;
        } else {
            int start = ip + GLZ_WIN;
            start = start > anchor + 3 ? start : anchor + 3;
            end = ip_bound;
            if (start < ip_bound) {
                end = start + glz_run_len<MODE>(base, start, glz_uniform(GLZ_W.tile[i + 1]),
                                                ip_bound - start, lane);
            }
        }
        len = end - anchor;
    } else {
        bool maxed = false, table = !(a & GLZ_A_P1V);
        int cand = 0;
        if (!table) {
            const int c1 = (int)((a >> GLZ_A_P1_SHIFT) & 255u);
            const bool st1 = (a & GLZ_A_P1SAME) ? (ins >> (c1 & 63)) & 1
                                                : (glz_readlane(vfl, c1 & 63) >> (c1 >> 6)) & 1;
            if (st1) {
                len = (int)((a >> GLZ_A_LEN1_SHIFT) & 31u);
                maxed = a & GLZ_A_MAX1;
                sel = GLZ_SEL_P1;
                cand = c1;
            } else if (!(a & GLZ_A_P2V)) {
                table = true;
            } else {
                const int c2 = (int)(b & 255u);
                const bool st2 = (a & GLZ_A_P2SAME) ? (ins >> (c2 & 63)) & 1
                                                    : (glz_readlane(vfl, c2 & 63) >> (c2 >> 6)) & 1;
                if (st2) {
                    len = (int)((b >> GLZ_B_LEN2_SHIFT) & 31u);
                    maxed = b & GLZ_B_MAX2;
                    sel = GLZ_SEL_P2;
                    cand = c2;
                } else if (!(a & GLZ_A_MORE)) {
                    table = true;
                } else {
                    /* both were skipped: look for an older store to the slot */
                    LONE_CLASS(1);
                    const uint32_t hi = glz_uniform(GLZ_W.hash[i]);
                    const uint64_t lo = (1ull << l) - 1;
                    int j = -1;
#pragma unroll
                    for (int w = 0; w < 4; w++) {
                        if (w <= Q) {
                            const uint64_t same = __ballot(GLZ_W.hash[w * 64 + lane] == hi);
                            const uint64_t stored = w < Q ? __ballot((vfl >> w) & 1u) : (ins & lo);
                            const uint64_t m = same & stored;
                            if (m) {
                                j = w * 64 + 63 - __clzll((long long)m);
                            }
                        }
                    }
                    if (j < 0) {
                        table = true;
                    } else {
                        const int maxn = ip_bound - anchor;
                        int checked = 0;
                        len = glz_tile_match_len<MODE>(GLZ_W.tile, i, j, maxn, lane, &checked);
                        if (len == checked && checked < maxn) {
                            len = glz_match_len<MODE>(base, anchor, base, ip + j, len, maxn, lane);
                        }
                        len = glz_accept_near<MODE>(len);
                        sel = GLZ_SEL_DEEP;
                        if (lane == 0) {
                            GLZ_W.tok_pdist[i] = (uint32_t)(i - j);
                        }
                    }
                }
            }
        }
        if (table) {
            len = (int)((a >> GLZ_A_GLEN_SHIFT) & 31u);
            maxed = a & GLZ_A_GMAX;
            sel = GLZ_SEL_TABLE;
            if (maxed) {
                LONE_CLASS(2);
                const uint32_t rseg = glz_uniform(GLZ_W.g_seg[i]);
                const GlzDevSeg rs = glz_seg(job, sh, rseg);
                const GLZ_GLOBAL uint8_t *rbase = glz_seg_base(job, rseg, rs);
                const int ref = (int)glz_uniform(GLZ_W.g_ref[i]);
                int maxn = ip_bound - anchor;
                const int rmax = (int)rs.nunits - ref;
                maxn = rmax < maxn ? rmax : maxn;
#ifdef GLZ_LONE
                /* measured ahead of the walk?  The match of a head is the match of every position
                 * behind it, up to its end, whose candidate lies at the same distance */
                bool known = false;
                const uint32_t hready = glz_uniform(*(volatile uint32_t *)&GLZ_W.heads_ready);
                __builtin_amdgcn_wave_barrier(); /* the records are read after the flags */
#pragma unroll
                for (int hd = 0; hd < GLZ_HEADS; hd++) {
                    const int hp = (int)glz_uniform((uint32_t)GLZ_W.head_pos[hd]);
                    const int he = (int)glz_uniform((uint32_t)GLZ_W.head_end[hd]);
                    if (!known && ((hready >> hd) & 1u) && hp >= 0 && hp <= i && glz_uniform(GLZ_W.head_seg[hd]) == rseg &&
                        (int)glz_uniform((uint32_t)GLZ_W.head_delta[hd]) == ref - anchor && he - anchor >= len) {
                        len = he - anchor;
                        known = glz_uniform((uint32_t)GLZ_W.head_done[hd]) != 0;
                    }
                }
                if (known) {
                    LONE_CLASS(4); /* measured ahead of the walk */
                }
                if (!known)
#endif
                len = glz_match_len<MODE>(base, anchor, rbase, ref, len, maxn, lane);
            }
        } else if (maxed) {
            /* a candidate inside the window: both sides are in the LDS tile as far
             * as it reaches, and most of these matches end a few pixels on */
            LONE_CLASS(3);
            const int maxn = ip_bound - anchor;
            int checked = 0;
            len = glz_tile_match_len<MODE>(GLZ_W.tile, i, cand, maxn, lane, &checked);
            if (len == checked && checked < maxn) {
                len = glz_match_len<MODE>(base, anchor, base, ip + cand, len, maxn, lane);
            }
        }
    }
    len = (int)glz_uniform((uint32_t)len); /* every lane holds the same value already */
    sel = glz_uniform(sel);
    if (lane == 0) {
        GLZ_W.tok[i] = (uint32_t)len | (sel << 29);
    }
    return len;
}

/* ---- resolve: wavefront 0 walks the stops of the window in order, a quarter
 * (64 positions) at a time so that the quarter's marks are 64-bit scalars.
 * Publishes per quarter: stops that became matches (ms) or literals (lit),
 * table stores (ins), and the stops whose outcome is in tok[] (slow). */
template <int MODE>
__device__ __forceinline__ void glz_window_resolve(GLZ_WIN_ARG const GlzDevJob &job, uint32_t seg_idx,
                                                   const GLZ_GLOBAL uint8_t *base, int ip, int npix,
                                                   int lane GLZ_ENTRY_ARGS)
{
    typedef GlzTraits<MODE> T;
    GlzShared &sh = glz_sh;
    const int ip_bound = npix - 2;
    const uint64_t q0 = glz_uniform64(GLZ_W.neq[0]), q1 = glz_uniform64(GLZ_W.neq[1]);
    const uint64_t q2 = glz_uniform64(GLZ_W.neq[2]), q3 = glz_uniform64(GLZ_W.neq[3]);
    uint32_t vfl = 0;
    int i = GLZ_ENTRY_I;
    int t0 = GLZ_ENTRY_T0, t1 = GLZ_ENTRY_T1; /* stores of the last match's tail that lie beyond its quarter */
    int qnext = 0;        /* first quarter whose marks are not published yet */
    bool open = true;

#ifdef GLZ_LONE_STAMPS
    const long long w_in = LONE_STAMP();
    long long w_acc = 0;
#endif
    while (open && i < GLZ_WIN) {
#ifdef GLZ_LONE_STAMPS
        const long long qs0 = LONE_STAMP();
#endif
        const int Q = i >> 6;
        /* quarters a match jumped over hold nothing but its tail's stores */
        for (int q = qnext; q < Q; q++) {
            const uint64_t m = glz_tail_bits(t0, t1, q);
            if (lane == 0) {
                GLZ_W.ins[q] = m;
            }
            vfl |= (uint32_t)((m >> lane) & 1) << q;
        }
        uint64_t ins = glz_tail_bits(t0, t1, Q), stops = 0, ms = 0, slow = 0;
        __builtin_amdgcn_wave_barrier(); /* lane 0's marks of the finished quarters are read below */
        uint32_t qa, qb; /* this lane's position: the scan's two result words */
        const uint32_t W = glz_quarter_words<MODE>(GLZ_WIN_PASS Q, lane, qa, qb);
        int l = i & 63;
        int last_len = 0;
#ifdef GLZ_LONE_STAMPS
        const long long qs1 = LONE_STAMP();
        LONE_ADD(0, qs1 - qs0);
        w_acc += qs1 - qs0;
#endif

        while (l < 64) {
            /* ordinary stops: everything needed is in the quarter word */
#ifdef GLZ_STAMPS
            const long long f0 = STAMP();
#endif
#ifdef GLZ_LONE_STAMPS
            const long long lf0 = LONE_STAMP();
#endif
            glz_fast_stops(W, l, ins, stops, ms, last_len);
#ifdef GLZ_STAMPS
            sh.st_fast += STAMP() - f0;
#endif
#ifdef GLZ_LONE_STAMPS
            if (lane == 0) glz_sh.st_fast += LONE_STAMP() - lf0;
#endif
            if (l >= 64) {
                break;
            }
            /* the others: no stop here at all, a run, a compare that goes on
             * past GLZ_K, a slot whose nearest two predecessors were skipped */
            const uint32_t sa = glz_readlane(qa, l), sb = glz_readlane(qb, l);
            if (!(sa & GLZ_A_ACT)) {
                open = false; /* the segment's last stops are behind us */
                break;
            }
            {
                const uint64_t bit = 1ull << l;
                bool run;
#ifdef GLZ_STAMPS
                const long long g0 = STAMP();
#endif
#ifdef GLZ_LONE_STAMPS
                const long long lg0 = LONE_STAMP();
#endif
                const int len = glz_resolve_general<MODE>(GLZ_WIN_PASS job, base, ip, ip_bound, Q * 64 + l, sa, sb, ins, vfl,
                                                          q0, q1, q2, q3, lane, &run);
#ifdef GLZ_STAMPS
                sh.st_gen += STAMP() - g0;
                sh.st_ngen++;
#endif
#ifdef GLZ_LONE_STAMPS
                if (lane == 0) {
                    glz_sh.st_gen[glz_sh.st_class] += LONE_STAMP() - lg0;
                    glz_sh.st_ngen[glz_sh.st_class]++;
                }
#endif
                slow |= bit;
                stops |= bit;
                ins |= run ? 0ull : bit;
                if (len) {
                    const uint64_t tm = (!T::GUARD_A || len > 2 ? 1ull : 0ull) | (!T::GUARD_B || len > 1 ? 2ull : 0ull);
                    const int sh0 = l + len - 2;
                    ms |= bit;
                    ins |= sh0 < 64 ? tm << sh0 : 0ull;
                }
                last_len = len;
                l += len ? len : 1;
            }
        }

#ifdef GLZ_LONE_STAMPS
        const long long qe0 = LONE_STAMP();
#endif
        i = Q * 64 + l;
        t0 = t1 = -1;
        if (open && last_len) {
            /* the part of the last match's tail that lies beyond this quarter */
            const int p0 = i - 2, p1 = i - 1;
            if ((!T::GUARD_A || last_len > 2) && p0 >= (Q + 1) * 64) {
                t0 = p0;
            }
            if ((!T::GUARD_B || last_len > 1) && p1 >= (Q + 1) * 64) {
                t1 = p1;
            }
        }
        if (lane == 0) {
            GLZ_W.ins[Q] = ins;
            GLZ_W.lit[Q] = stops & ~ms;
            GLZ_W.ms[Q] = ms;
            GLZ_W.slow[Q] = slow;
        }
        vfl |= (uint32_t)((ins >> lane) & 1) << Q;
        qnext = Q + 1;
#ifdef GLZ_LONE_STAMPS
        const long long qe1 = LONE_STAMP();
        LONE_ADD(1, qe1 - qe0);
        LONE_ADD(3, 1);
#endif
    }
    /* quarters behind the last one walked: only tail stores can be in them */
    for (int q = qnext; q < 4; q++) {
        const uint64_t m = glz_tail_bits(t0, t1, q);
        if (lane == 0) {
            GLZ_W.ins[q] = m;
        }
    }

    /* Stores beyond the window (glz_encode_tmpl.c:424-444) wait for the next
     * window: its tile holds the pixels their hashes are made of, so they cost no
     * memory round trip here (glz_parse_seg makes them). */
#ifdef GLZ_LONE_STAMPS
    LONE_ADD(2, LONE_STAMP() - w_in);
#endif
    if (lane == 0) {
        GLZ_W.next_ip = ip + i; /* i counts every pixel consumed, also past the window */
        GLZ_W.pend = (t0 >= GLZ_WIN ? 1u : 0u) | (t1 >= GLZ_WIN ? 2u : 0u);
#ifdef GLZ_LONE
        GLZ_W.tail0 = t0;
        GLZ_W.tail1 = t1;
#endif
    }
}

/* ---- emit: every lane writes the token of its own window position, and the
 * window's table stores go out (the last store to a slot wins) */
template <int MODE>
__device__ __forceinline__ void glz_window_emit(GLZ_WIN_ARG const GlzDevJob &job, uint32_t seg_idx, int ip,
                                                int tid, uint32_t my_h, bool my_hv GLZ_EMIT_ARG)
{
    typedef GlzTraits<MODE> T;
    GlzShared &sh = glz_sh;
#ifdef GLZ_STAMPS
    long long es_ = STAMP();
#endif
    GLZ_RECOMPUTE(tid);
    const int lane = tid & 63, p = tid;
    const int w = (int)glz_uniform((uint32_t)(tid >> 6));
    const uint64_t lit0 = glz_uniform64(GLZ_W.lit[0]), lit1 = glz_uniform64(GLZ_W.lit[1]);
    const uint64_t lit2 = glz_uniform64(GLZ_W.lit[2]), lit3 = glz_uniform64(GLZ_W.lit[3]);
    const uint64_t ms0 = glz_uniform64(GLZ_W.ms[0]), ms1 = glz_uniform64(GLZ_W.ms[1]);
    const uint64_t ms2 = glz_uniform64(GLZ_W.ms[2]), ms3 = glz_uniform64(GLZ_W.ms[3]);
    const uint64_t lit[4] = {lit0, lit1, lit2, lit3};
    const uint64_t ms[4] = {ms0, ms1, ms2, ms3};
    const uint32_t op_in = glz_uniform(sh.op), c0 = glz_uniform(sh.copy), lc_in = glz_uniform(sh.last_copy);
    const uint64_t below = (1ull << lane) - 1;

    /* (1) table stores: a stored position writes its slot unless a later store
     * of this window (or of its last match's tail) goes to the same slot */
#ifdef GLZ_LONE
    if (stores)
#endif
    {
        const uint32_t *ins32 = (const uint32_t *)GLZ_W.ins;
        if (my_hv && ((ins32[p >> 5] >> (p & 31)) & 1u)) {
            /* the last match's stores beyond the window are made after these, in
             * program order, so they win a shared slot by themselves */
            bool dominated = GLZ_DOMINATED_AT_ONCE;
            const uint32_t key = my_h & (GLZ_KEYS - 1);
            for (int v = p >> 5; v < 8 && !dominated; v++) {
                uint32_t bits = GLZ_W.bitmap[GLZ_BM(key, (uint32_t)v)] & ins32[v];
                if (v == (p >> 5)) {
                    bits &= ~((2u << (p & 31)) - 1);
                }
                while (bits && !dominated) {
                    const int b = __ffs((int)bits) - 1;
                    bits &= bits - 1;
                    dominated = GLZ_W.hash[v * 32 + b] == my_h;
                }
            }
            if (!dominated) {
                glz_htab_store(job, my_h, seg_idx, (uint32_t)(ip + p));
            }
        }
    }

#ifdef GLZ_LONE
    if (stores) {
        return; /* the tokens go out a step later, beside the next window's walk */
    }
#endif
    EMIT_STAMP(0);
    /* (2) where this wavefront's 64 positions stand in the literal runs */
    uint32_t carry = c0;     /* literals of the open run before this wavefront's positions */
    bool nomatch = true;     /* ... and no match starts before them in the window */
    uint32_t after = 0;      /* literals behind them up to the next match start */
    bool closes = false;     /* ... if there is one */
    uint32_t run_end = c0;   /* literals of the run that stays open at the window's end */
#pragma unroll
    for (int v = 0; v < 4; v++) {
        if (ms[v]) {
            const int lm = 63 - __clzll((long long)ms[v]);
            const uint32_t tail = (uint32_t)__popcll((lit[v] >> lm) >> 1);
            if (v < w) {
                carry = tail;
                nomatch = false;
            }
            run_end = tail;
        } else {
            const uint32_t n = (uint32_t)__popcll(lit[v]);
            if (v < w) {
                carry += n;
            }
            run_end += n;
        }
        if (v > w && !closes) {
            if (ms[v]) {
                const int fm = __ffsll((unsigned long long)ms[v]) - 1;
                after += (uint32_t)__popcll(lit[v] & ((1ull << fm) - 1));
                closes = true;
            } else {
                after += (uint32_t)__popcll(lit[v]);
            }
        }
    }
    const uint64_t litw = glz_pick4(lit0, lit1, lit2, lit3, w);
    const uint64_t msw = glz_pick4(ms0, ms1, ms2, ms3, w);
    const bool is_lit = (litw >> lane) & 1, is_ms = (msw >> lane) & 1;

    /* r: literals of the current run before this position */
    uint32_t r;
    bool first_run; /* the run reaches back beyond the window */
    {
        const uint64_t msb = msw & below, litb = litw & below;
        if (msb) {
            const int lm = 63 - __clzll((long long)msb);
            r = (uint32_t)__popcll((litb >> lm) >> 1);
            first_run = false;
        } else {
            r = carry + (uint32_t)__popcll(litb);
            first_run = nomatch;
        }
    }
    /* rem: literals of the run from this position on; closed: a match ends it in this window */
    uint32_t rem;
    bool closed;
    {
        const uint64_t msa = (msw >> lane) >> 1, lita = litw >> lane;
        if (msa) {
            const int d = __ffsll((unsigned long long)msa) - 1;
            rem = (uint32_t)__popcll(lita & ((2ull << d) - 1));
            closed = true;
        } else {
            rem = (uint32_t)__popcll(lita) + after;
            closed = closes;
        }
    }

    EMIT_STAMP(1);
    uint32_t bytes = 0, elen = 0, idist = 0, pdist = 0, nff = 0;
    bool cs = false;
    if (is_lit) {
        cs = (r & 31) == 0;
        bytes = T::LITB + (cs ? 1u : 0u);
    } else if (is_ms) {
        const uint32_t *ins32 = (const uint32_t *)GLZ_W.ins;
        const uint32_t a = GLZ_W.infoA[p], b = GLZ_W.infoB[p];
        const uint32_t n1 = (a >> GLZ_A_P1_SHIFT) & 255u, n2 = b & 255u;
        /* Everything any of the cases below could want is asked for at once (the addresses
         * are in range whatever the flags say; what a case does not use is stale data that is
         * never looked at) and the choice is a chain of selects: two LDS round trips instead
         * of one per nested test.  Same precedence as the walk's: a stop it resolved itself,
         * a run, the nearest stored position with this hash, the second nearest, the table. */
        const uint32_t w1 = ins32[n1 >> 5], w2 = ins32[n2 >> 5];
        const uint32_t slow_tok = GLZ_W.tok[p], gi = GLZ_W.g_idist[p], gp = GLZ_W.g_pdist[p], tp = GLZ_W.tok_pdist[p];
        const bool slow = (GLZ_W.slow[p >> 6] >> lane) & 1;
        const bool s1 = (a & GLZ_A_P1V) && ((w1 >> (n1 & 31u)) & 1u);
        const bool s2 = (a & GLZ_A_P2V) && ((w2 >> (n2 & 31u)) & 1u);
        uint32_t tok = ((a >> GLZ_A_GLEN_SHIFT) & 31u) | (GLZ_SEL_TABLE << 29);
        tok = s2 ? ((b >> GLZ_B_LEN2_SHIFT) & 31u) | (GLZ_SEL_P2 << 29) : tok;
        tok = s1 ? ((a >> GLZ_A_LEN1_SHIFT) & 31u) | (GLZ_SEL_P1 << 29) : tok;
        tok = (a & GLZ_A_RLE) ? (b & 0x1ffu) | (GLZ_SEL_RUN << 29) : tok;
        tok = slow ? slow_tok : tok;
        const uint32_t sel = tok >> 29;
        pdist = sel == GLZ_SEL_P1 ? (uint32_t)p - n1
              : sel == GLZ_SEL_P2 ? (uint32_t)p - n2
              : sel == GLZ_SEL_TABLE ? gp
              : sel == GLZ_SEL_RUN ? 1u : tp;
        idist = sel == GLZ_SEL_TABLE ? gi : 0u;
        if (idist == 0) {
            pdist--; /* glz_encode_tmpl.c:389-391 */
        }
        elen = (tok & GLZ_TOK_LEN_MASK) - T::LEN_BIAS;
        bytes = glz_match_bytes(idist, pdist, elen);
        nff = elen >= 7 ? (elen - 7) / 255u : 0;
    }

    EMIT_STAMP(2);
    /* (3) byte offsets: prefix sum over the 256 positions */
    const uint32_t incl = wave_scan_inclusive(bytes, lane); /* six DPP adds, no LDS round trips */
    if (lane == 63) {
        sh.wave_sum[w] = incl;
    }
    GLZ_PHASE_BARRIER(1);
    const uint32_t ws0 = sh.wave_sum[0], ws1 = sh.wave_sum[1], ws2 = sh.wave_sum[2], ws3 = sh.wave_sum[3];
    uint32_t off = op_in + incl - bytes + (w > 0 ? ws0 : 0u) + (w > 1 ? ws1 : 0u) + (w > 2 ? ws2 : 0u);
    /* the cursor moves on here, while the four sums are at hand: held until the end of the
     * phase they are four registers the allocator spills and reloads in every window
     * (every lane read the old cursor before the barrier above) */
    if (tid == 0) {
        sh.op = op_in + ws0 + ws1 + ws2 + ws3;
        sh.copy = run_end & 31;
    }

    EMIT_STAMP(3);
    /* (4) write.  A token that does not fit whole is not written at all: the image
     * then ends in GLZ_DEV_ERR_NO_SPACE (op runs past out_cap) and nobody reads its bytes. */
    const bool fits = off + bytes <= job.out_cap;
    if (is_lit) {
        if (cs) {
            if (fits) {
                glz_put(glz_out(job), GLZ_NO_CAP, off, closed ? (rem < 32 ? rem : 32) - 1 : GLZ_MAX_COPY - 1);
            }
            if (!closed && rem <= 32) {
                sh.last_copy = off; /* the count byte a later window may have to patch */
            }
            off++;
        }
        if (fits) {
            glz_put_pixel<MODE>(glz_out(job), GLZ_NO_CAP, off, GLZ_W.tile[p + 1]);
        }
    } else if (is_ms) {
        if (first_run && c0 > 0 && r < 32) {
            glz_put(glz_out(job), job.out_cap, lc_in, r - 1); /* glz_encoder.c:95-108 */
        }
        if (fits) {
            glz_write_match(glz_out(job), GLZ_NO_CAP, off, idist, pdist, elen, 16);
        }
    }
    {
        uint64_t big = __ballot(is_ms && nff > 16);
        while (big) {
            const int src = __ffsll((unsigned long long)big) - 1;
            big &= big - 1;
            const uint32_t o2 = __shfl(off, src), n2 = __shfl(nff, src);
            for (uint32_t t = (uint32_t)lane; t < n2; t += GLZ_WAVE) {
                glz_put(glz_out(job), job.out_cap, o2 + 1 + t, 255u);
            }
        }
    }
    /* the bitmap is clean again for the next window */
    if (my_hv) {
        GLZ_W.bitmap[GLZ_BM(my_h & (GLZ_KEYS - 1), (uint32_t)p >> 5)] = 0;
    }
    EMIT_STAMP(4);
}

#ifndef GLZ_LONE
/* One segment, glz_encode_tmpl.c:268-491.  All 256 lanes call this.
 * from: first position to parse (2 in an image's first segment, whose two
 * leading pixels are already out as literals). */
template <int MODE>
__device__ __forceinline__ void glz_parse_seg(const GlzDevJob &job, uint32_t seg_idx, int from, int tid)
{
    GlzShared &sh = glz_sh;
    const int lane = tid & 63;
    const bool w0 = tid < GLZ_WAVE;
    const GlzDevSeg cs = glz_seg_uniform(glz_seg(job, sh, seg_idx));
    const GLZ_GLOBAL uint8_t *base = glz_seg_base(job, seg_idx, cs);
    const int npix = (int)cs.nunits;
    const int ip_limit = npix - 6;
    int ip = from;
    if (tid == 0) {
        GLZ_W.pend = 0;
    }
    __syncthreads();

#ifdef GLZ_STAMPS
    long long a_tile = 0, a_scan = 0, a_res = 0, a_emit = 0, a_win = 0;
    if (tid == 0) {
        sh.st_fast = sh.st_gen = sh.st_ngen = 0;
        for (int i = 0; i < 8; i++) sh.st_scan[i] = sh.st_emit[i] = 0;
    }
#endif
    while (ip < ip_limit) {
#ifdef GLZ_STAMPS
        const long long t0 = STAMP();
#endif
        for (int t = tid; t < GLZ_TILE; t += GLZ_THREADS) {
            const int pos = ip - 1 + t;
            GLZ_W.tile[t] = (pos >= 0 && pos < npix) ? glz_load_px<MODE>(base, pos) : 0u;
        }
        if (tid == GLZ_THREADS - 1) {
            GLZ_W.tile_m2 = ip >= 2 ? glz_load_px<MODE>(base, ip - 2) : 0u;
        }
        __syncthreads();
#ifdef GLZ_STAMPS
        const long long t1 = STAMP();
#endif
        /* the table stores the previous window's last match left behind, now that
         * the pixels are here: position ip-2 first, then ip-1, as the reference
         * orders them */
        const uint32_t pend = glz_uniform(GLZ_W.pend);
        uint32_t th0 = GLZ_NO_HASH, th1 = GLZ_NO_HASH;
        if (pend) {
            typedef GlzTraits<MODE> T;
            const uint32_t a = GLZ_W.tile_m2 & T::MASK, b = GLZ_W.tile[0] & T::MASK;
            const uint32_t c = GLZ_W.tile[1] & T::MASK, d = GLZ_W.tile[2] & T::MASK;
            th0 = glz_hash3<MODE>(a, b, c);
            th1 = glz_hash3<MODE>(b, c, d);
            if (tid == 0) {
                if (pend & 1u) {
                    glz_htab_store(job, th0, seg_idx, (uint32_t)(ip - 2));
                }
                if (pend & 2u) {
                    glz_htab_store(job, th1, seg_idx, (uint32_t)(ip - 1));
                }
            }
        }
        uint32_t my_h;
        bool my_hv;
        glz_window_scan<MODE>(job, cs, seg_idx, base, ip, npix, tid, pend, th0, th1, my_h, my_hv);
        __syncthreads();
#ifdef GLZ_STAMPS
        const long long t2 = STAMP();
#endif
        if (w0) {
            /* The walk is the one phase a session cannot spread over lanes: while it runs,
             * the workgroup's other three wavefronts wait at the barrier and the session's
             * progress is this wavefront's instruction issue.  On a full GPU it competes
             * for issue slots with the four-wavefront phases of the CU's other seven
             * sessions, which have lanes to spare; it goes first (measured: a step of 3072
             * sessions 72.8 -> 71.2 ms; priorities for the other phases as well, 1 or 2
             * below this one, gave 71.6 - 72.1). */
            SPICE_SETPRIO(3);
            glz_window_resolve<MODE>(job, seg_idx, base, ip, npix, lane);
            SPICE_SETPRIO(0);
        }
        __syncthreads();
#ifdef GLZ_STAMPS
        const long long t3 = STAMP();
#endif
        glz_window_emit<MODE>(job, seg_idx, ip, tid, my_h, my_hv);
        __syncthreads();
#ifdef GLZ_STAMPS
        const long long t4 = STAMP();
        a_tile += t1 - t0; a_scan += t2 - t1; a_res += t3 - t2; a_emit += t4 - t3; a_win++;
#endif
        ip = GLZ_W.next_ip;
    }
#ifdef GLZ_STAMPS
    if (tid == 0) {
        atomicAdd(&glz_stamp_acc[0], (unsigned long long)a_tile);
        atomicAdd(&glz_stamp_acc[1], (unsigned long long)a_scan);
        atomicAdd(&glz_stamp_acc[2], (unsigned long long)a_res);
        atomicAdd(&glz_stamp_acc[3], (unsigned long long)a_emit);
        atomicAdd(&glz_stamp_acc[4], (unsigned long long)a_win);
        atomicAdd(&glz_stamp_acc[5], (unsigned long long)sh.st_fast);
        atomicAdd(&glz_stamp_acc[6], (unsigned long long)sh.st_gen);
        atomicAdd(&glz_stamp_acc[7], (unsigned long long)sh.st_ngen);
        for (int i = 0; i < 6; i++) atomicAdd(&glz_stamp_acc[8 + i], (unsigned long long)sh.st_scan[i]);
        for (int i = 0; i < 5; i++) atomicAdd(&glz_stamp_acc[14 + i], (unsigned long long)sh.st_emit[i]);
    }
#endif

    /* ---- leftovers as literals, then the run closes; glz_encode_tmpl.c:466-487 */
    if (w0) {
        /* stores the last window's last match left behind: no tile follows */
        const uint32_t pend = glz_uniform(GLZ_W.pend);
        if (pend & 1u) {
            const uint32_t th0 = glz_hash_at<MODE>(base, ip - 2);
            if (lane == 0) {
                glz_htab_store(job, th0, seg_idx, (uint32_t)(ip - 2));
            }
        }
        if (pend & 2u) {
            const uint32_t th1 = glz_hash_at<MODE>(base, ip - 1);
            if (lane == 0) {
                glz_htab_store(job, th1, seg_idx, (uint32_t)(ip - 1));
            }
        }
        if (lane == 0) {
            GLZ_W.pend = 0;
        }
        glz_run_from_arena<MODE>(job, base, ip, npix - ip, true, lane);
    }
    __syncthreads();
}

#else
#include "glz_lone_seg.h"
#endif

/* glz_encode_tmpl.c:498-555 */
template <int MODE>
__device__ __forceinline__ void glz_parse_image(const GlzDevJob &job, const uint32_t *seg_list, int tid)
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
            const GLZ_GLOBAL uint8_t *base = glz_seg_base(job, seg_list[s], cs);
            glz_run_from_arena<MODE>(job, base, 0, (int)cs.nunits, true, lane);
        }
        __syncthreads();
    }
    bool first = true;
    for (; s < job.nsegs; s++) {
        const uint32_t sid = seg_list[s];
        if (first) {
            /* the first real segment starts with a hash insert and two literals */
            if (w0) {
                const GlzDevSeg cs = glz_seg(job, sh, sid);
                const GLZ_GLOBAL uint8_t *base = glz_seg_base(job, sid, cs);
                if (lane == 0) {
                    glz_htab_store(job, glz_hash_at<MODE>(base, 0), sid, 0);
                }
                glz_run_from_arena<MODE>(job, base, 0, 2, false, lane);
            }
            __syncthreads();
        }
        glz_parse_seg<MODE>(job, sid, first ? 2 : 0, tid);
        first = false;
    }
}

