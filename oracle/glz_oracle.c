/*
 * TEST INFRASTRUCTURE ONLY -- see glz_oracle.h for scope and parity status.
 *
 * CPU restatement of:
 *   server/glz_encoder.c              (byte sink, header, per-type dispatch)
 *   server/glz_encoder_dictionary.c   (window of images/segments, eviction)
 *   server/glz_encoder_dictionary_protected.h (hash slot, reference validity)
 *   server/glz_encode_match_tmpl.c    (reference encoding and its size)
 *   server/glz_encode_tmpl.c          (the parser; in glz_oracle_parse.inc)
 *   client/glz_decode_tmpl.c + client/glz_decoder.cpp (decoder, for round trips)
 */
#include "glz_oracle.h"

#include <stdlib.h>
#include <string.h>

#define GLZ_HASH_LOG 20                       /* glz_encoder_dictionary_protected.h:34 */
#define GLZ_HASH_SIZE (1u << GLZ_HASH_LOG)
#define GLZ_HASH_MASK (GLZ_HASH_SIZE - 1)
#define GLZ_NULL_SEG 0xffffffffu              /* glz_encoder_dictionary_protected.h:55-56 */
#define GLZ_INIT_SEGS 1000                    /* glz_encoder_dictionary_protected.h:57 */
#define GLZ_MAX_COPY 32                       /* client/glz_decode_tmpl.c:193 */
#define GLZ_MAX_WINDOW (1u << 25)             /* client/red_client.cpp:875 */
#define GLZ_BOUND_OFFSET 2                    /* glz_encoder.c:191 */
#define GLZ_LIMIT_OFFSET 6                    /* glz_encoder.c:192 */
#define GLZ_MAX_PIXEL_SHORT_DISTANCE 4096u    /* glz_encoder.c:195-198 */
#define GLZ_MAX_PIXEL_MEDIUM_DISTANCE 131072u
#define GLZ_MAX_PIXEL_LONG_DISTANCE 33554432u
#define GLZ_MAX_IMAGE_DIST 16777215u

static const int TYPE_IS_PLT[12] = {0, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0};
static const int TYPE_IS_RGB[12] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1};
static const int TYPE_PLT_PPB[12] = {0, 8, 8, 2, 2, 1, 0, 0, 0, 0, 0, 0};
static const int TYPE_RGB_BPP[12] = {0, 1, 1, 1, 1, 1, 2, 3, 4, 4, 4, 1};

typedef struct OImage OImage;
struct OImage {
    uint64_t id;
    int type;
    int size; /* pixels */
    uint32_t first_seg;
    uint64_t usr_tag;
    OImage *next;
    uint8_t alive;
};

typedef struct OSeg {
    OImage *image;
    uint8_t *lines;
    uint8_t *lines_end;
    uint8_t *owned; /* oracle-side copy of the caller's rows */
    uint32_t pixels_num;
    uint64_t pixels_so_far;
    uint32_t next;
} OSeg;

typedef struct OHashEntry {
    uint32_t seg_idx;
    uint32_t pix_idx;
} OHashEntry;

struct GlzoDict {
    OSeg *segs;
    uint32_t segs_quota;
    uint32_t used_segs_head;
    uint32_t used_segs_tail;
    uint32_t free_segs_head;
    uint32_t *encoders_heads;
    OImage *used_images_tail;
    OImage *used_images_head;
    OImage *free_images;
    uint64_t pixels_so_far;
    uint32_t size_limit;
    OHashEntry *htab;
    uint64_t last_image_id;
    uint32_t max_encoders;

    uint64_t *freed;
    int n_freed, cap_freed;
    GlzoStats stats;
};

typedef struct OEnc {
    GlzoDict *dict;
    uint32_t id;
    int type;
    uint64_t image_id;
    uint32_t first_seg;
    uint8_t *out;
    size_t pos, cap;
    size_t last_copy;
    int overflow;
} OEnc;

/* Optional trace of the parser's stops (seg-relative anchor, length; 0 = literal,
 * bit 31 = run).  Analysis aid for tests; not thread safe. */
static uint32_t *g_trace;
static size_t g_trace_cap, g_trace_n;
void glzo_set_trace(uint32_t *buf, size_t cap_pairs)
{
    g_trace = buf;
    g_trace_cap = cap_pairs;
    g_trace_n = 0;
}
size_t glzo_trace_count(void)
{
    return g_trace_n;
}
#define TRACE(a, l)                                                    \
    do {                                                               \
        if (g_trace && g_trace_n < g_trace_cap) {                      \
            g_trace[2 * g_trace_n] = (uint32_t)(a);                    \
            g_trace[2 * g_trace_n + 1] = (uint32_t)(l);                \
            g_trace_n++;                                               \
        }                                                              \
    } while (0)

/* ---------------------------------------------------------------- byte sink
 * glz_encoder.c:62-108.  The reference writes into a chain of buffers; the
 * stream it produces is the same as one linear buffer. */
static void put_byte(OEnc *e, uint8_t b)
{
    if (e->pos >= e->cap) {
        e->overflow = 1;
        e->pos++;
        return;
    }
    e->out[e->pos++] = b;
}

static void put_u32(OEnc *e, uint32_t w)
{
    put_byte(e, (uint8_t)(w >> 24));
    put_byte(e, (uint8_t)(w >> 16));
    put_byte(e, (uint8_t)(w >> 8));
    put_byte(e, (uint8_t)w);
}

/* glz_encoder.c:83-87 -- only 24 bits of the low word survive */
static void put_u64(OEnc *e, uint64_t w)
{
    put_u32(e, (uint32_t)(w >> 32));
    put_u32(e, (uint32_t)(w & 0xffffff));
}

static void put_copy_count(OEnc *e, uint8_t c)
{
    put_byte(e, c);
    e->last_copy = e->pos - 1;
}

static void patch_copy_count(OEnc *e, uint8_t c)
{
    if (e->last_copy < e->cap) {
        e->out[e->last_copy] = c;
    }
}

static void drop_last_byte(OEnc *e)
{
    e->pos--;
}

/* ------------------------------------------------- reference (match) coding
 * glz_encode_match_tmpl.c:31-151 */
static int ref_encoding_bytes(uint32_t image_dist, size_t pix_dist)
{
    int n;
    if (pix_dist < GLZ_MAX_PIXEL_SHORT_DISTANCE) {
        if (image_dist < (1u << 6)) {
            n = 3;
        } else if (image_dist < (1u << 14)) {
            n = 4;
        } else if (image_dist < (1u << 22)) {
            n = 5;
        } else {
            n = 6;
        }
    } else {
        if (image_dist == 0) {
            n = 3;
        } else if (image_dist < (1u << 8)) {
            n = 4;
        } else if (image_dist < (1u << 16)) {
            n = 5;
        } else {
            n = 6;
        }
        if (pix_dist >= GLZ_MAX_PIXEL_MEDIUM_DISTANCE) {
            n++;
        }
    }
    return n;
}

static void put_match(OEnc *e, uint32_t image_dist, size_t pix_dist, size_t len)
{
    int far_px = pix_dist >= GLZ_MAX_PIXEL_SHORT_DISTANCE;
    uint8_t low = (uint8_t)((far_px ? 16 : 0) + (pix_dist & 0x0f));

    if (len < 7) {
        put_byte(e, (uint8_t)((len << 5) + low));
    } else {
        put_byte(e, (uint8_t)((7 << 5) + low));
        for (len -= 7; len >= 255; len -= 255) {
            put_byte(e, 255);
        }
        put_byte(e, (uint8_t)len);
    }
    put_byte(e, (uint8_t)((pix_dist >> 4) & 255));

    if (!far_px) {
        if (image_dist < (1u << 6)) {
            put_byte(e, (uint8_t)(image_dist & 0x3f));
        } else if (image_dist < (1u << 14)) {
            put_byte(e, (uint8_t)((1 << 6) + (image_dist & 0x3f)));
            put_byte(e, (uint8_t)((image_dist >> 6) & 255));
        } else if (image_dist < (1u << 22)) {
            put_byte(e, (uint8_t)((1 << 7) + (image_dist & 0x3f)));
            put_byte(e, (uint8_t)((image_dist >> 6) & 255));
            put_byte(e, (uint8_t)((image_dist >> 14) & 255));
        } else {
            put_byte(e, (uint8_t)((1 << 7) + (1 << 6) + (image_dist & 0x3f)));
            put_byte(e, (uint8_t)((image_dist >> 6) & 255));
            put_byte(e, (uint8_t)((image_dist >> 14) & 255));
            put_byte(e, (uint8_t)((image_dist >> 22) & 255));
        }
    } else {
        uint8_t very_far = (pix_dist < GLZ_MAX_PIXEL_MEDIUM_DISTANCE) ? 0 : 32;
        uint8_t hi = (uint8_t)((pix_dist >> 12) & 31);
        if (image_dist == 0) {
            put_byte(e, (uint8_t)(very_far + hi));
        } else if (image_dist < (1u << 8)) {
            put_byte(e, (uint8_t)(very_far + (1 << 6) + hi));
            put_byte(e, (uint8_t)(image_dist & 255));
        } else if (image_dist < (1u << 16)) {
            put_byte(e, (uint8_t)(very_far + (1 << 7) + hi));
            put_byte(e, (uint8_t)(image_dist & 255));
            put_byte(e, (uint8_t)((image_dist >> 8) & 255));
        } else {
            put_byte(e, (uint8_t)(very_far + (1 << 7) + (1 << 6) + hi));
            put_byte(e, (uint8_t)(image_dist & 255));
            put_byte(e, (uint8_t)((image_dist >> 8) & 255));
            put_byte(e, (uint8_t)((image_dist >> 16) & 255));
        }
        if (very_far) {
            put_byte(e, (uint8_t)((pix_dist >> 17) & 255));
        }
    }
}

/* glz_encoder_dictionary_protected.h:176-184 */
static int ref_seg_usable(GlzoDict *d, uint32_t enc_id, OSeg *ref, OSeg *src)
{
    if (ref == src) {
        return 1;
    }
    return ref->image && ref->image->alive && src->image->type == ref->image->type &&
           ref->pixels_so_far <= src->pixels_so_far &&
           d->segs[d->encoders_heads[enc_id]].pixels_so_far <= ref->pixels_so_far;
}

/* glz_encode_tmpl.c:21-22 */
#define DJB_STEP(h, c) ((h) = (((h) << 5) + (h)) ^ (uint32_t)(c))

static inline uint32_t hash_bytes3(uint8_t a, uint8_t b, uint8_t c)
{
    uint32_t h = 5381;
    DJB_STEP(h, a);
    DJB_STEP(h, b);
    DJB_STEP(h, c);
    return h & GLZ_HASH_MASK;
}

/* three b,g,r pixels hashed in r,g,b order, glz_encode_tmpl.c:110-122 */
static inline uint32_t hash_rgb(const uint8_t *p, int stride)
{
    uint32_t h = 5381;
    int i;
    for (i = 0; i < 3; i++) {
        DJB_STEP(h, p[2]);
        DJB_STEP(h, p[1]);
        DJB_STEP(h, p[0]);
        p += stride;
    }
    return h & GLZ_HASH_MASK;
}

/* glz_encode_tmpl.c:77-86 */
static inline uint32_t hash_rgb16(const uint8_t *p)
{
    uint32_t h = 5381;
    int i;
    for (i = 0; i < 3; i++) {
        uint16_t v = (uint16_t)(p[0] | (p[1] << 8));
        DJB_STEP(h, v & 0x00ff);
        DJB_STEP(h, (v >> 8) & 0x007f);
        p += 2;
    }
    return h & GLZ_HASH_MASK;
}

/* b,g,r compared, pad ignored (glz_encode_tmpl.c:125-128) */
static inline int same_rgb32(const uint8_t *a, const uint8_t *b)
{
    uint32_t x, y;
    memcpy(&x, a, 4);
    memcpy(&y, b, 4);
    return ((x ^ y) & 0x00ffffffu) == 0;
}

static inline int same_rgb16(const uint8_t *a, const uint8_t *b)
{
    uint16_t x = (uint16_t)(a[0] | (a[1] << 8)), y = (uint16_t)(b[0] | (b[1] << 8));
    return ((x ^ y) & 0x7fff) == 0;
}

/* ---- palette: one byte per "pixel" (glz_encode_tmpl.c:33-48) */
#define TAG plt
#define PXB 1
#define SAME(a, b) ((a)[0] == (b)[0])
#define EMIT(e, p) put_byte(e, (p)[0])
#define HASH3(p) hash_bytes3((p)[0], (p)[1], (p)[2])
#define MIN_REF 4
#define MAX_REF 7
#define ENC_DIV 1
#define LEN_BIAS 2
#define GUARD_A 0
#define GUARD_B 0
#define PPB(e) (TYPE_PLT_PPB[(e)->type])
#include "glz_oracle_parse.inc"

/* ---- alpha plane of an RGBA image (glz_encode_tmpl.c:50-65) */
#define TAG alpha
#define PXB 4
#define SAME(a, b) ((a)[3] == (b)[3])
#define EMIT(e, p) put_byte(e, (p)[3])
#define HASH3(p) hash_bytes3((p)[3], (p)[7], (p)[11])
#define MIN_REF 4
#define MAX_REF 7
#define ENC_DIV 1
#define LEN_BIAS 2
#define GUARD_A 0
#define GUARD_B 0
#define PPB(e) 1
#include "glz_oracle_parse.inc"

/* ---- x1r5g5b5 (glz_encode_tmpl.c:68-87) */
#define TAG rgb16
#define PXB 2
#define SAME(a, b) same_rgb16(a, b)
#define EMIT(e, p) do { put_byte(e, (p)[1]); put_byte(e, (p)[0]); } while (0)
#define HASH3(p) hash_rgb16(p)
#define MIN_REF 2
#define MAX_REF 3
#define ENC_DIV 2
#define LEN_BIAS 1
#define GUARD_A 1
#define GUARD_B 0
#define PPB(e) 1
#include "glz_oracle_parse.inc"

/* ---- b,g,r (glz_encode_tmpl.c:89-95) */
#define TAG rgb24
#define PXB 3
#define SAME(a, b) ((a)[0] == (b)[0] && (a)[1] == (b)[1] && (a)[2] == (b)[2])
#define EMIT(e, p) do { put_byte(e, (p)[0]); put_byte(e, (p)[1]); put_byte(e, (p)[2]); } while (0)
#define HASH3(p) hash_rgb(p, 3)
#define MIN_REF 2
#define MAX_REF 2
#define ENC_DIV 3
#define LEN_BIAS 0
#define GUARD_A 1
#define GUARD_B 1
#define PPB(e) 1
#include "glz_oracle_parse.inc"

/* ---- b,g,r,pad (glz_encode_tmpl.c:97-103) */
#define TAG rgb32
#define PXB 4
#define SAME(a, b) same_rgb32(a, b)
#define EMIT(e, p) do { put_byte(e, (p)[0]); put_byte(e, (p)[1]); put_byte(e, (p)[2]); } while (0)
#define HASH3(p) hash_rgb(p, 4)
#define MIN_REF 2
#define MAX_REF 2
#define ENC_DIV 3
#define LEN_BIAS 0
#define GUARD_A 1
#define GUARD_B 1
#define PPB(e) 1
#include "glz_oracle_parse.inc"

/* ------------------------------------------------------------- dictionary */

static void log_freed(GlzoDict *d, uint64_t tag)
{
    if (d->n_freed == d->cap_freed) {
        d->cap_freed = d->cap_freed ? d->cap_freed * 2 : 64;
        d->freed = (uint64_t *)realloc(d->freed, sizeof(uint64_t) * (size_t)d->cap_freed);
    }
    d->freed[d->n_freed++] = tag;
}

static void seg_blank(OSeg *s, uint32_t next)
{
    s->image = NULL;
    s->lines = NULL;
    s->lines_end = NULL;
    s->pixels_num = 0;
    s->pixels_so_far = 0;
    s->next = next;
}

/* glz_encoder_dictionary.c:30-45 */
static void release_all_images(GlzoDict *d)
{
    while (d->used_images_head) {
        OImage *im = d->used_images_head;
        d->used_images_head = im->next;
        if (im->alive) {
            log_freed(d, im->usr_tag);
        }
        im->next = d->free_images;
        im->alive = 0;
        d->free_images = im;
    }
    d->used_images_tail = NULL;
}

/* glz_encoder_dictionary.c:82-117,203-212 */
void glzo_dict_reset(GlzoDict *d)
{
    uint32_t i;
    d->last_image_id = 0;
    d->free_segs_head = 0;
    for (i = 0; i < d->segs_quota; i++) {
        free(d->segs[i].owned);
        d->segs[i].owned = NULL;
        seg_blank(&d->segs[i], i + 1);
    }
    d->segs[d->segs_quota - 1].next = GLZ_NULL_SEG;
    d->used_segs_head = GLZ_NULL_SEG;
    d->used_segs_tail = GLZ_NULL_SEG;
    for (i = 0; i < d->max_encoders; i++) {
        d->encoders_heads[i] = GLZ_NULL_SEG;
    }
    release_all_images(d);
    memset(d->htab, 0, sizeof(OHashEntry) * GLZ_HASH_SIZE);
}

/* glz_encoder_dictionary.c:48-78,147-176 */
GlzoDict *glzo_dict_create(uint32_t size_px, uint32_t max_encoders)
{
    GlzoDict *d;
    if (size_px > GLZ_MAX_WINDOW || max_encoders == 0) {
        return NULL;
    }
    d = (GlzoDict *)calloc(1, sizeof(*d));
    d->max_encoders = max_encoders;
    d->size_limit = size_px;
    d->segs_quota = GLZ_INIT_SEGS;
    d->segs = (OSeg *)calloc(d->segs_quota, sizeof(OSeg));
    d->encoders_heads = (uint32_t *)calloc(max_encoders, sizeof(uint32_t));
    d->htab = (OHashEntry *)calloc(GLZ_HASH_SIZE, sizeof(OHashEntry));
    d->pixels_so_far = 0;
    glzo_dict_reset(d);
    return d;
}

/* glz_encoder_dictionary.c:191-201 */
GlzoDict *glzo_dict_restore(uint32_t size_px, uint32_t max_encoders, uint64_t last_image_id)
{
    GlzoDict *d = glzo_dict_create(size_px, max_encoders);
    if (d) {
        d->last_image_id = last_image_id;
    }
    return d;
}

/* glz_encoder_dictionary.c:178-189 */
void glzo_dict_get_restore_data(GlzoDict *d, uint32_t *size_px, uint32_t *max_encoders,
                                uint64_t *last_image_id)
{
    *size_px = d->size_limit;
    *max_encoders = d->max_encoders;
    *last_image_id = d->last_image_id;
}

uint32_t glzo_dict_get_size(GlzoDict *d)
{
    return d ? d->size_limit : 0;
}

void glzo_dict_destroy(GlzoDict *d)
{
    uint32_t i;
    if (!d) {
        return;
    }
    release_all_images(d);
    for (i = 0; i < d->segs_quota; i++) {
        free(d->segs[i].owned);
    }
    free(d->segs);
    while (d->free_images) {
        OImage *im = d->free_images;
        d->free_images = im->next;
        free(im);
    }
    free(d->encoders_heads);
    free(d->htab);
    free(d->freed);
    free(d);
}

/* glz_encoder_dictionary.c:242-252 -- logical removal, no callback */
void glzo_remove_image(GlzoDict *d, void *image)
{
    (void)d;
    ((OImage *)image)->alive = 0;
}

int glzo_pop_freed(GlzoDict *d, uint64_t *tags, int max)
{
    int n = d->n_freed < max ? d->n_freed : max;
    memcpy(tags, d->freed, sizeof(uint64_t) * (size_t)n);
    memmove(d->freed, d->freed + n, sizeof(uint64_t) * (size_t)(d->n_freed - n));
    d->n_freed -= n;
    return n;
}

void glzo_get_stats(GlzoDict *d, GlzoStats *out, int reset)
{
    *out = d->stats;
    if (reset) {
        memset(&d->stats, 0, sizeof(d->stats));
    }
}

/* glz_encoder_dictionary.c:267-309 */
static void grow_segs(GlzoDict *d)
{
    uint32_t old = d->segs_quota, quota = old * 2, i;
    d->segs = (OSeg *)realloc(d->segs, sizeof(OSeg) * quota);
    for (i = old; i < quota; i++) {
        d->segs[i].owned = NULL;
        seg_blank(&d->segs[i], i + 1);
    }
    d->segs[quota - 1].next = d->free_segs_head;
    d->free_segs_head = old;
    d->segs_quota = quota;
}

/* glz_encoder_dictionary.c:312-336 */
static OImage *take_image(GlzoDict *d)
{
    OImage *im;
    if (d->free_images) {
        im = d->free_images;
        d->free_images = im->next;
    } else {
        im = (OImage *)calloc(1, sizeof(*im));
    }
    im->next = NULL;
    if (d->used_images_tail) {
        d->used_images_tail->next = im;
    }
    d->used_images_tail = im;
    if (!d->used_images_head) {
        d->used_images_head = im;
    }
    return im;
}

/* glz_encoder_dictionary.c:339-356 */
static uint32_t take_seg(GlzoDict *d)
{
    uint32_t id;
    if (d->free_segs_head == GLZ_NULL_SEG) {
        grow_segs(d);
    }
    id = d->free_segs_head;
    d->free_segs_head = d->segs[id].next;
    return id;
}

/* glz_encoder_dictionary.c:392-424 */
static OImage *head_after_adding(GlzoDict *d, int new_size)
{
    uint32_t win;
    OImage *head;
    if (!d->used_images_head) {
        return NULL;
    }
    head = d->segs[d->used_segs_head].image;
    win = d->segs[d->used_segs_tail].pixels_num +
          (uint32_t)(d->segs[d->used_segs_tail].pixels_so_far -
                     d->segs[d->used_segs_head].pixels_so_far);
    while (win + (uint32_t)new_size > d->size_limit) {
        win -= (uint32_t)head->size;
        head = head->next;
    }
    return head;
}

static int dict_busy(GlzoDict *d)
{
    uint32_t i;
    for (i = 0; i < d->max_encoders; i++) {
        if (d->encoders_heads[i] != GLZ_NULL_SEG) {
            return 1;
        }
    }
    return 0;
}

/* glz_encoder_dictionary.c:359-387,439-459 */
static void drop_head_until(GlzoDict *d, OImage *end_image)
{
    while (d->used_images_head != end_image) {
        OImage *im = d->used_images_head;
        uint32_t old_free = d->free_segs_head, sid, nxt;

        d->free_segs_head = im->first_seg;
        for (sid = im->first_seg, nxt = d->segs[sid].next;
             nxt != GLZ_NULL_SEG && d->segs[nxt].image == im;
             sid = nxt, nxt = d->segs[sid].next) {
        }
        d->segs[sid].next = old_free;

        d->used_images_head = im->next;
        if (im->alive) {
            log_freed(d, im->usr_tag);
        }
        im->alive = 0;
        im->next = d->free_images;
        d->free_images = im;
    }
    if (!d->used_images_head) {
        d->used_segs_head = GLZ_NULL_SEG;
        d->used_segs_tail = GLZ_NULL_SEG;
        d->used_images_tail = NULL;
    } else {
        d->used_segs_head = end_image->first_seg;
    }
}

/* glz_encoder_dictionary.c:461-548 */
static OImage *append_image(GlzoDict *d, int type, int image_size, int height, int stride,
                            const uint8_t *const *chunks, const uint32_t *chunk_rows, int nchunks,
                            uint64_t usr_tag)
{
    OImage *im = take_image(d);
    uint32_t sid = GLZ_NULL_SEG, prev = GLZ_NULL_SEG, row = 0;
    int c;

    im->id = d->last_image_id++;
    im->size = image_size;
    im->type = type;
    im->usr_tag = usr_tag;

    for (c = 0; c < nchunks; c++) {
        uint32_t rows = chunk_rows[c];
        size_t bytes = (size_t)rows * (size_t)stride;
        OSeg *s;
        sid = take_seg(d);
        s = &d->segs[sid];
        free(s->owned);
        s->owned = (uint8_t *)malloc(bytes ? bytes : 1);
        memcpy(s->owned, chunks[c], bytes);
        s->image = im;
        s->lines = s->owned;
        s->lines_end = s->owned + bytes;
        s->pixels_num = (uint32_t)((int64_t)image_size * rows / height);
        s->pixels_so_far = d->pixels_so_far;
        d->pixels_so_far += s->pixels_num;
        s->next = GLZ_NULL_SEG;
        if (row == 0) {
            im->first_seg = sid;
        } else {
            d->segs[prev].next = sid;
        }
        row += rows;
        prev = sid;
        if (row >= (uint32_t)height) {
            break;
        }
    }

    if (d->used_segs_tail == GLZ_NULL_SEG) {
        d->used_segs_head = im->first_seg;
        d->used_segs_tail = sid;
    } else {
        d->segs[d->used_segs_tail].next = im->first_seg;
        d->used_segs_tail = sid;
    }
    im->alive = 1;
    return im;
}

/* glz_encoder_dictionary.c:593-633 */
static int seg_earlier(GlzoDict *d, uint32_t dst, uint32_t src)
{
    return src == GLZ_NULL_SEG ||
           (dst != GLZ_NULL_SEG && d->segs[dst].pixels_so_far < d->segs[src].pixels_so_far);
}

static void after_encode(GlzoDict *d, uint32_t enc_id)
{
    uint32_t i, early = GLZ_NULL_SEG, mine;
    for (i = 0; i < d->max_encoders; i++) {
        if (i != enc_id && seg_earlier(d, d->encoders_heads[i], early)) {
            early = d->encoders_heads[i];
        }
    }
    if (seg_earlier(d, d->used_segs_head, early)) {
        early = d->used_segs_head;
    }
    mine = d->encoders_heads[enc_id];
    if (seg_earlier(d, mine, early)) {
        drop_head_until(d, d->segs[early].image);
    }
    d->encoders_heads[enc_id] = GLZ_NULL_SEG;
}

/* glz_encoder.c:225-310 with glz_encoder_dictionary.c:550-591 folded in */
long glzo_encode(GlzoDict *d, uint32_t encoder_id, int type, int width, int height,
                 int top_down, int stride, const uint8_t *const *chunks,
                 const uint32_t *chunk_rows, int nchunks, uint8_t *out, size_t out_cap,
                 uint64_t usr_tag, void **o_image)
{
    OEnc e;
    OImage *im, *new_head;
    int image_size, c;
    uint32_t rows_total = 0, head_dist;

    if (type <= GLZO_TYPE_INVALID || type > GLZO_TYPE_RGBA) {
        return GLZO_ERR_BAD_TYPE;
    }
    if (TYPE_IS_PLT[type]) {
        int min_stride = width / TYPE_PLT_PPB[type];
        if (stride > min_stride &&
            ((width % TYPE_PLT_PPB[type]) == 0 || (stride - min_stride) > 1)) {
            return GLZO_ERR_STRIDE;
        }
    } else if (stride != width * TYPE_RGB_BPP[type]) {
        return GLZO_ERR_STRIDE;
    }
    for (c = 0; c < nchunks; c++) {
        rows_total += chunk_rows[c];
    }
    if (nchunks <= 0 || rows_total < (uint32_t)height) {
        return GLZO_ERR_LINES;
    }

    /* glz_encoder_dictionary.c:258-265 */
    if (TYPE_IS_RGB[type]) {
        image_size = height * stride / TYPE_RGB_BPP[type];
    } else {
        image_size = height * stride * TYPE_PLT_PPB[type];
    }
    if ((uint32_t)image_size > d->size_limit) {
        return GLZO_ERR_TOO_BIG;
    }

    new_head = head_after_adding(d, image_size);
    if (!dict_busy(d)) {
        drop_head_until(d, new_head);
    }
    im = append_image(d, type, image_size, height, stride, chunks, chunk_rows, nchunks, usr_tag);
    if (new_head) {
        d->encoders_heads[encoder_id] = new_head->first_seg;
        head_dist = (uint32_t)(im->id - new_head->id);
    } else {
        d->encoders_heads[encoder_id] = im->first_seg;
        head_dist = 0;
    }
    if (o_image) {
        *o_image = im;
    }

    memset(&e, 0, sizeof(e));
    e.dict = d;
    e.id = encoder_id;
    e.type = type;
    e.image_id = im->id;
    e.first_seg = im->first_seg;
    e.out = out;
    e.cap = out_cap;

    put_byte(&e, 0x20); /* magic: "LZ  " loaded little-endian, written big-endian */
    put_byte(&e, 0x20);
    put_byte(&e, 0x5a);
    put_byte(&e, 0x4c);
    put_u32(&e, 0x00010001u);
    put_byte(&e, (uint8_t)((type & 0x0f) | (top_down ? (1 << 4) : 0)));
    put_u32(&e, (uint32_t)width);
    put_u32(&e, (uint32_t)height);
    put_u32(&e, (uint32_t)stride);
    put_u64(&e, im->id);
    put_u32(&e, head_dist);

    switch (type) {
    case GLZO_TYPE_PLT1_BE:
    case GLZO_TYPE_PLT1_LE:
    case GLZO_TYPE_PLT4_BE:
    case GLZO_TYPE_PLT4_LE:
    case GLZO_TYPE_PLT8:
        parse_image_plt(&e);
        break;
    case GLZO_TYPE_RGB16:
        parse_image_rgb16(&e);
        break;
    case GLZO_TYPE_RGB24:
        parse_image_rgb24(&e);
        break;
    case GLZO_TYPE_RGB32:
        parse_image_rgb32(&e);
        break;
    case GLZO_TYPE_RGBA:
        parse_image_rgb32(&e);
        parse_image_alpha(&e);
        break;
    default:
        break;
    }

    after_encode(d, encoder_id);
    if (e.overflow) {
        return GLZO_ERR_NO_SPACE;
    }
    return (long)e.pos;
}

/* ------------------------------------------------------------------ decoder
 * client/glz_decoder.cpp:51-105,238-301 and client/glz_decode_tmpl.c:179-319,
 * decoding to the image's own pixel type (DECODE_TO_SAME). */

typedef struct DImage {
    uint64_t id;
    int type;
    uint8_t *data;
    size_t gross_pixels;
    int bpp;
} DImage;

struct GlzoDecWindow {
    DImage *imgs;
    int n, cap;
};

GlzoDecWindow *glzo_decwin_create(void)
{
    return (GlzoDecWindow *)calloc(1, sizeof(GlzoDecWindow));
}

void glzo_decwin_destroy(GlzoDecWindow *w)
{
    int i;
    if (!w) {
        return;
    }
    for (i = 0; i < w->n; i++) {
        free(w->imgs[i].data);
    }
    free(w->imgs);
    free(w);
}

static DImage *decwin_find(GlzoDecWindow *w, uint64_t id)
{
    int i;
    for (i = w->n - 1; i >= 0; i--) {
        if (w->imgs[i].id == id) {
            return &w->imgs[i];
        }
    }
    return NULL;
}

static void decwin_trim(GlzoDecWindow *w, uint64_t oldest)
{
    int i, k = 0;
    for (i = 0; i < w->n; i++) {
        if (w->imgs[i].id < oldest) {
            free(w->imgs[i].data);
        } else {
            w->imgs[k++] = w->imgs[i];
        }
    }
    w->n = k;
}

static uint32_t rd_u32(const uint8_t *p)
{
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
}

/* mode: 0 whole pixel of bpp bytes from `lit` literal bytes; 1 alpha plane */
static long decode_plane(GlzoDecWindow *w, const uint8_t *in, size_t in_len, uint8_t *out,
                         size_t npix, int type, uint64_t id, int alpha)
{
    const uint8_t *ip = in, *in_end = in + in_len;
    int bpp = TYPE_RGB_BPP[type];
    int len_bias = alpha ? 2 : (TYPE_IS_PLT[type] ? 2 : (type == GLZO_TYPE_RGB16 ? 1 : 0));
    size_t op = 0;
    uint32_t ctrl;

    if (npix == 0) {
        return 0;
    }
    if (ip >= in_end) {
        return GLZO_ERR_DECODE;
    }
    ctrl = *ip++;
    for (;;) {
        if (ctrl >= GLZ_MAX_COPY) {
            uint32_t len = ctrl >> 5;
            uint32_t far_px = (ctrl >> 4) & 1;
            uint32_t pix_ofs = ctrl & 0x0f;
            uint32_t image_dist, flag, code;
            int i;
            const uint8_t *ref;

            if (len == 7) {
                do {
                    if (ip >= in_end) return GLZO_ERR_DECODE;
                    code = *ip++;
                    len += code;
                } while (code == 255);
            }
            if (ip + 2 > in_end) return GLZO_ERR_DECODE;
            code = *ip++;
            pix_ofs += code << 4;
            code = *ip++;
            flag = (code >> 6) & 3;
            if (!far_px) {
                image_dist = code & 0x3f;
                for (i = 0; i < (int)flag; i++) {
                    if (ip >= in_end) return GLZO_ERR_DECODE;
                    image_dist += (uint32_t)(*ip++) << (6 + 8 * i);
                }
            } else {
                uint32_t very_far = (code >> 5) & 1;
                pix_ofs += (code & 0x1f) << 12;
                image_dist = 0;
                for (i = 0; i < (int)flag; i++) {
                    if (ip >= in_end) return GLZO_ERR_DECODE;
                    image_dist += (uint32_t)(*ip++) << (8 * i);
                }
                if (very_far) {
                    if (ip >= in_end) return GLZO_ERR_DECODE;
                    pix_ofs += (uint32_t)(*ip++) << 17;
                }
            }
            len += (uint32_t)len_bias;
            if (!image_dist) {
                pix_ofs += 1;
            }
            if (op + len > npix) {
                return GLZO_ERR_DECODE;
            }
            if (!image_dist) {
                if (pix_ofs > op) {
                    return GLZO_ERR_DECODE;
                }
                ref = out + (op - pix_ofs) * (size_t)bpp;
            } else {
                DImage *ri = decwin_find(w, id - image_dist);
                if (!ri || ri->type != type || (size_t)pix_ofs + len > ri->gross_pixels) {
                    return GLZO_ERR_DECODE;
                }
                ref = ri->data + (size_t)pix_ofs * (size_t)bpp;
            }
            /* pixel-by-pixel so that overlapping (run) references replicate */
            for (; len; len--) {
                if (alpha) {
                    out[op * 4 + 3] = ref[3];
                } else {
                    int b;
                    for (b = 0; b < bpp; b++) {
                        out[op * (size_t)bpp + (size_t)b] = ref[b];
                    }
                }
                ref += bpp;
                op++;
            }
        } else {
            uint32_t n = ctrl + 1;
            if (op + n > npix) {
                return GLZO_ERR_DECODE;
            }
            for (; n; n--) {
                uint8_t *o = out + op * (size_t)bpp;
                if (alpha) {
                    if (ip >= in_end) return GLZO_ERR_DECODE;
                    o[3] = *ip++;
                } else if (type == GLZO_TYPE_RGB32 || type == GLZO_TYPE_RGBA) {
                    if (ip + 3 > in_end) return GLZO_ERR_DECODE;
                    o[0] = ip[0];
                    o[1] = ip[1];
                    o[2] = ip[2];
                    o[3] = 0;
                    ip += 3;
                } else if (type == GLZO_TYPE_RGB24) {
                    if (ip + 3 > in_end) return GLZO_ERR_DECODE;
                    o[0] = ip[0];
                    o[1] = ip[1];
                    o[2] = ip[2];
                    ip += 3;
                } else if (type == GLZO_TYPE_RGB16) {
                    if (ip + 2 > in_end) return GLZO_ERR_DECODE;
                    o[1] = ip[0];
                    o[0] = ip[1];
                    ip += 2;
                } else {
                    if (ip >= in_end) return GLZO_ERR_DECODE;
                    o[0] = *ip++;
                }
                op++;
            }
        }
        if (op < npix) {
            if (ip >= in_end) return GLZO_ERR_DECODE;
            ctrl = *ip++;
        } else {
            break;
        }
    }
    return (long)(ip - in);
}

long glzo_decode(GlzoDecWindow *w, const uint8_t *in, size_t in_len, uint8_t *out,
                 size_t out_cap, int *o_type, int *o_width, int *o_height, int *o_top_down,
                 uint64_t *o_id, uint32_t *o_win_head_dist)
{
    int type, top_down, bpp;
    uint32_t width, height, stride, head_dist;
    uint64_t id;
    size_t gross, units;
    long used, used2 = 0;
    DImage *slot;

    if (in_len < 33 || rd_u32(in) != 0x20205a4cu || rd_u32(in + 4) != 0x00010001u) {
        return GLZO_ERR_DECODE;
    }
    type = in[8] & 0x0f;
    top_down = (in[8] >> 4) ? 1 : 0;
    width = rd_u32(in + 9);
    height = rd_u32(in + 13);
    stride = rd_u32(in + 17);
    id = ((uint64_t)rd_u32(in + 21) << 32) | rd_u32(in + 25);
    head_dist = rd_u32(in + 29);
    if (type <= GLZO_TYPE_INVALID || type > GLZO_TYPE_RGBA) {
        return GLZO_ERR_DECODE;
    }
    if (TYPE_IS_PLT[type]) {
        gross = (size_t)stride * (size_t)TYPE_PLT_PPB[type] * height;
        units = gross / (size_t)TYPE_PLT_PPB[type];
    } else {
        gross = (size_t)width * height;
        units = gross;
    }
    bpp = TYPE_RGB_BPP[type];
    if (units * (size_t)bpp > out_cap) {
        return GLZO_ERR_DECODE;
    }
    decwin_trim(w, id - head_dist);

    used = decode_plane(w, in + 33, in_len - 33, out, units, type, id, 0);
    if (used < 0) {
        return used;
    }
    if (type == GLZO_TYPE_RGBA) {
        used2 = decode_plane(w, in + 33 + used, in_len - 33 - (size_t)used, out, units, type, id, 1);
        if (used2 < 0) {
            return used2;
        }
    }

    if (w->n == w->cap) {
        w->cap = w->cap ? w->cap * 2 : 16;
        w->imgs = (DImage *)realloc(w->imgs, sizeof(DImage) * (size_t)w->cap);
    }
    slot = &w->imgs[w->n++];
    slot->id = id;
    slot->type = type;
    slot->gross_pixels = units;
    slot->bpp = bpp;
    slot->data = (uint8_t *)malloc(units * (size_t)bpp + 1);
    memcpy(slot->data, out, units * (size_t)bpp);

    if (o_type) *o_type = type;
    if (o_width) *o_width = (int)width;
    if (o_height) *o_height = (int)height;
    if (o_top_down) *o_top_down = top_down;
    if (o_id) *o_id = id;
    if (o_win_head_dist) *o_win_head_dist = head_dist;
    return 33 + used + used2;
}
