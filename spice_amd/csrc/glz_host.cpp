/*
 * GLZ: host bookkeeping of the shared dictionary and the C entry points.
 *
 * The window of images/segments is pure control state (a few words per image)
 * and stays on the host, following server/glz_encoder_dictionary.c step by
 * step so that segment slots, image ids and eviction points are the ones the
 * reference would produce -- the hash table in HBM stores segment slot numbers,
 * and a recycled slot is a legitimate match source (the reference never clears
 * the table on eviction), so slot reuse order is part of the bitstream.
 * Pixels, the hash table and the device view of the segment table live in HBM.
 *
 * Encodes on one dictionary are serialised by its mutex; with that, a
 * dictionary is never "in use" by another encoder when an image is added
 * (glz_encoder_dictionary.c:569-571), which is the only regime in which the
 * reference's output is defined independently of thread timing.
 */
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "glz_device.h"
#include "runtime.h"

using namespace spicegpu;

void glz_launch_parse(const GlzDevJob *jobs, int njobs, unsigned type_mask, const uint32_t *u32_pool,
                      const GlzDevSegUpdate *upd_pool, hipStream_t stream);
void glz_launch_parse_lone(const GlzDevJob *jobs, int njobs, unsigned type_mask, const uint32_t *u32_pool,
                           const GlzDevSegUpdate *upd_pool, hipStream_t stream, int skip_blocks);
void glz_launch_copy(const GlzCopyDesc *descs, int ndescs, unsigned blocks_per_desc,
                     hipStream_t stream);

namespace {

const uint32_t kNullSeg = 0xffffffffu;      /* glz_encoder_dictionary_protected.h:55-56 */
const uint32_t kInitSegs = 1000;            /* glz_encoder_dictionary_protected.h:57 */
const uint32_t kMaxWindow = 1u << 25;       /* LZ_MAX_WINDOW_SIZE, client/red_client.cpp:875 */

const int kTypeIsRgb[12] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1};
const int kTypeIsPlt[12] = {0, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0};
const int kPltPixelsPerByte[12] = {0, 8, 8, 2, 2, 1, 0, 0, 0, 0, 0, 0};
const int kRgbBytesPerPixel[12] = {0, 1, 1, 1, 1, 1, 2, 3, 4, 4, 4, 1};

struct HostImage {
    uint64_t id;
    int type;
    int size; /* pixels */
    uint32_t first_seg;
    GlzUsrImageContext *usr_context;
    HostImage *next;
    bool alive;
    uint64_t arena_begin, arena_end;
};

struct HostSeg {
    HostImage *image;
    uint64_t arena_off;
    uint32_t nunits;
    uint32_t pixels_num;
    uint64_t pixels_so_far;
    uint32_t next;
};

struct Dict {
    std::recursive_mutex lock; /* free_image callbacks may call back into the dictionary */
    uint32_t size_limit = 0;
    uint32_t max_encoders = 0;
    uint64_t last_image_id = 0;
    uint64_t pixels_so_far = 0;
    std::vector<HostSeg> segs;
    uint32_t used_segs_head = kNullSeg, used_segs_tail = kNullSeg, free_segs_head = kNullSeg;
    std::vector<uint32_t> encoders_heads;
    HostImage *used_images_head = nullptr, *used_images_tail = nullptr, *free_images = nullptr;
    GlzEncoderUsrContext *cur_usr = nullptr;
    /* compressed bytes per raw byte of this dictionary's last batch image; 2 until
     * one exists, so that new sessions count as expensive.  Orders a batch. */
    double cost_hint = 2.0;
    /* a batch between submit and finish has images of this dictionary on the device (set
     * and cleared with the dictionary locked) */
    bool in_flight = false;

    /* HBM side */
    uint8_t *d_arena = nullptr;
    uint64_t arena_cap = 0;
    uint2 *d_htab = nullptr;
    GlzDevSeg *d_segs = nullptr;
    uint32_t d_segs_cap = 0;
    std::vector<uint32_t> dirty; /* segment slots whose device copy is stale */

    /* FIFO ring allocator over the arena */
    uint64_t ring_head = 0, ring_tail = 0;
    bool ring_wrapped = false;
    uint64_t largest_run = 0; /* most arena bytes one image has taken so far */
    uint32_t arena_regrows = 0;
};

struct Encoder {
    GlzEncoderUsrContext *usr;
    uint8_t id;
    Dict *dict;
    /* the synchronous path (glz_encode): a stream of its own, so that worker
     * threads with different dictionaries run side by side on the device */
    hipStream_t st = nullptr;
    Staging staging;     /* job record and result */
    DevBuffer out;       /* compressed stream */
    PinnedBuffer bounce; /* its host landing zone */
};

uint64_t align16(uint64_t v)
{
    return (v + 15) & ~15ull;
}

/* ------------------------------------------------------------ ring allocator
 * Images leave the window oldest first, so their pixel runs form a FIFO. */
bool ring_alloc(Dict *d, uint64_t bytes, uint64_t *at)
{
    if (!d->used_images_head) {
        d->ring_head = d->ring_tail = 0;
        d->ring_wrapped = false;
    }
    if (!d->ring_wrapped) {
        if (d->arena_cap - d->ring_tail >= bytes) {
            *at = d->ring_tail;
            d->ring_tail += bytes;
            return true;
        }
        if (d->ring_head >= bytes) {
            *at = 0;
            d->ring_tail = bytes;
            d->ring_wrapped = true;
            return true;
        }
        return false;
    }
    if (d->ring_head - d->ring_tail >= bytes) {
        *at = d->ring_tail;
        d->ring_tail += bytes;
        return true;
    }
    return false;
}

void ring_after_free(Dict *d, uint64_t freed_begin)
{
    if (!d->used_images_head) {
        d->ring_head = d->ring_tail = 0;
        d->ring_wrapped = false;
        return;
    }
    uint64_t nh = d->used_images_head->arena_begin;
    if (d->ring_wrapped && nh < freed_begin) {
        d->ring_wrapped = false;
    }
    d->ring_head = nh;
}

uint64_t arena_slack(uint32_t window_px)
{
    /* 16-byte alignment of every segment, with a floor for small windows */
    uint64_t s = (uint64_t)window_px * 4 / 16;
    return s < (64u << 10) ? (64u << 10) : s;
}

void mark_image_dirty(Dict *d, HostImage *im);

/* A FIFO of contiguous runs wastes at most one run where it wraps, so an arena
 * of the window's bytes plus the largest run always has room for the next
 * image once the images it displaces are gone.  The arena starts at the
 * window's size and is re-made to that bound the first time an image does not
 * fit: live images move to the front of the new arena in window order (their
 * segments' offsets change and are uploaded again with the next job). */
bool arena_regrow(Dict *d, uint64_t need, hipStream_t st)
{
    uint64_t live = 0;
    for (HostImage *im = d->used_images_head; im; im = im->next) {
        live += im->arena_end - im->arena_begin;
        uint64_t run = im->arena_end - im->arena_begin;
        d->largest_run = run > d->largest_run ? run : d->largest_run;
    }
    d->largest_run = need > d->largest_run ? need : d->largest_run;
    uint64_t cap = (uint64_t)d->size_limit * 4 + d->largest_run + arena_slack(d->size_limit);
    if (cap < live + 2 * d->largest_run) {
        cap = live + 2 * d->largest_run; /* eviction held back by another encoder */
    }
    uint8_t *fresh = nullptr;
    if (!check(hipMalloc((void **)&fresh, cap + 64), "hipMalloc arena")) {
        return false;
    }
    uint64_t pos = 0;
    bool ok = true;
    for (HostImage *im = d->used_images_head; im && ok; im = im->next) {
        const uint64_t len = im->arena_end - im->arena_begin;
        ok = check(hipMemcpyAsync(fresh + pos, d->d_arena + im->arena_begin, len,
                                  hipMemcpyDeviceToDevice, st), "move window image");
        pos += len;
    }
    ok = ok && check(hipStreamSynchronize(st), "sync");
    if (!ok) {
        hipFree(fresh); /* nothing of the dictionary has changed yet */
        return false;
    }
    pos = 0;
    for (HostImage *im = d->used_images_head; im; im = im->next) {
        const uint64_t len = im->arena_end - im->arena_begin;
        for (uint32_t sid = im->first_seg; sid != kNullSeg && d->segs[sid].image == im;
             sid = d->segs[sid].next) {
            d->segs[sid].arena_off = d->segs[sid].arena_off - im->arena_begin + pos;
        }
        mark_image_dirty(d, im);
        im->arena_begin = pos;
        im->arena_end = pos + len;
        pos += len;
    }
    hipFree(d->d_arena);
    d->d_arena = fresh;
    d->arena_cap = cap;
    d->ring_head = 0;
    d->ring_tail = pos;
    d->ring_wrapped = false;
    d->arena_regrows++;
    return true;
}

/* -------------------------------------------------- device view of a segment */
GlzDevSeg dev_view(const Dict *d, const HostSeg &s)
{
    GlzDevSeg v;
    memset(&v, 0, sizeof(v));
    v.pixels_so_far = s.pixels_so_far;
    v.arena_off16 = (uint32_t)(s.arena_off >> 4);
    v.nunits = s.nunits;
    if (s.image) {
        const HostSeg &first = d->segs[s.image->first_seg];
        v.image_id = s.image->id;
        v.img_first_delta = (uint32_t)(s.pixels_so_far - first.pixels_so_far);
        v.flags = (uint32_t)s.image->type | GLZ_SEG_USED | (s.image->alive ? GLZ_SEG_ALIVE : 0u);
    }
    return v;
}

void mark_image_dirty(Dict *d, HostImage *im)
{
    for (uint32_t sid = im->first_seg; sid != kNullSeg && d->segs[sid].image == im;
         sid = d->segs[sid].next) {
        d->dirty.push_back(sid);
    }
}

/* ------------------------------------------------------------ window upkeep */
void blank_seg(HostSeg &s, uint32_t next)
{
    s.image = nullptr;
    s.arena_off = 0;
    s.nunits = 0;
    s.pixels_num = 0;
    s.pixels_so_far = 0;
    s.next = next;
}

/* glz_encoder_dictionary.c:30-45 */
void release_all_images(Dict *d)
{
    while (d->used_images_head) {
        HostImage *im = d->used_images_head;
        d->used_images_head = im->next;
        if (im->alive && d->cur_usr && d->cur_usr->free_image) {
            d->cur_usr->free_image(d->cur_usr, im->usr_context);
        }
        im->next = d->free_images;
        im->alive = false;
        d->free_images = im;
    }
    d->used_images_tail = nullptr;
}

bool ensure_dev_segs(Dict *d, hipStream_t st)
{
    uint32_t need = (uint32_t)d->segs.size();
    if (need <= d->d_segs_cap) {
        return true;
    }
    GlzDevSeg *n = nullptr;
    if (!check(hipMalloc((void **)&n, sizeof(GlzDevSeg) * (size_t)need), "hipMalloc segs")) {
        return false;
    }
    bool ok = check(hipMemsetAsync(n, 0, sizeof(GlzDevSeg) * (size_t)need, st), "memset segs");
    if (ok && d->d_segs) {
        ok = check(hipMemcpyAsync(n, d->d_segs, sizeof(GlzDevSeg) * (size_t)d->d_segs_cap,
                                  hipMemcpyDeviceToDevice, st), "copy segs");
    }
    ok = ok && check(hipStreamSynchronize(st), "sync");
    if (!ok) {
        hipFree(n);
        return false;
    }
    if (d->d_segs) {
        hipFree(d->d_segs);
    }
    d->d_segs = n;
    d->d_segs_cap = need;
    return true;
}

/* glz_encoder_dictionary.c:82-117,203-212 */
bool dict_reset(Dict *d)
{
    d->last_image_id = 0;
    d->free_segs_head = 0;
    for (uint32_t i = 0; i < d->segs.size(); i++) {
        blank_seg(d->segs[i], i + 1);
    }
    d->segs.back().next = kNullSeg;
    d->used_segs_head = d->used_segs_tail = kNullSeg;
    for (auto &h : d->encoders_heads) {
        h = kNullSeg;
    }
    release_all_images(d);
    d->dirty.clear();
    d->ring_head = d->ring_tail = 0;
    d->ring_wrapped = false;
    hipStream_t st = stream();
    GateHold dev("glz dictionary reset");
    if (!dev.ok) {
        return false;
    }
    return check(hipMemsetAsync(d->d_htab, 0, sizeof(uint2) * (size_t)GLZ_HASH_SIZE, st), "memset htab") &&
           check(hipMemsetAsync(d->d_segs, 0, sizeof(GlzDevSeg) * (size_t)d->d_segs_cap, st), "memset segs") &&
           check(hipStreamSynchronize(st), "sync");
}

void dict_free_device(Dict *d)
{
    if (d->d_arena) hipFree(d->d_arena);
    if (d->d_htab) hipFree(d->d_htab);
    if (d->d_segs) hipFree(d->d_segs);
    d->d_arena = nullptr;
    d->d_htab = nullptr;
    d->d_segs = nullptr;
}

/* glz_encoder_dictionary.c:267-309 */
void grow_segs(Dict *d)
{
    uint32_t old = (uint32_t)d->segs.size(), quota = old * 2;
    d->segs.resize(quota);
    for (uint32_t i = old; i < quota; i++) {
        blank_seg(d->segs[i], i + 1);
    }
    d->segs[quota - 1].next = d->free_segs_head;
    d->free_segs_head = old;
}

/* glz_encoder_dictionary.c:312-336 */
HostImage *take_image(Dict *d)
{
    HostImage *im;
    if (d->free_images) {
        im = d->free_images;
        d->free_images = im->next;
    } else {
        im = new HostImage();
    }
    im->next = nullptr;
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
uint32_t take_seg(Dict *d)
{
    if (d->free_segs_head == kNullSeg) {
        grow_segs(d);
    }
    uint32_t id = d->free_segs_head;
    d->free_segs_head = d->segs[id].next;
    return id;
}

/* glz_encoder_dictionary.c:392-424 */
HostImage *head_after_adding(Dict *d, int new_size)
{
    if (!d->used_images_head) {
        return nullptr;
    }
    HostImage *head = d->segs[d->used_segs_head].image;
    uint32_t win = d->segs[d->used_segs_tail].pixels_num +
                   (uint32_t)(d->segs[d->used_segs_tail].pixels_so_far -
                              d->segs[d->used_segs_head].pixels_so_far);
    while (win + (uint32_t)new_size > d->size_limit) {
        win -= (uint32_t)head->size;
        head = head->next;
    }
    return head;
}

bool dict_in_use(Dict *d)
{
    for (uint32_t h : d->encoders_heads) {
        if (h != kNullSeg) {
            return true;
        }
    }
    return false;
}

/* glz_encoder_dictionary.c:359-387,439-459 */
void drop_head_until(Dict *d, HostImage *end_image)
{
    while (d->used_images_head != end_image) {
        HostImage *im = d->used_images_head;
        uint32_t old_free = d->free_segs_head, sid, nxt;
        d->free_segs_head = im->first_seg;
        for (sid = im->first_seg, nxt = d->segs[sid].next;
             nxt != kNullSeg && d->segs[nxt].image == im; sid = nxt, nxt = d->segs[sid].next) {
        }
        d->segs[sid].next = old_free;
        d->used_images_head = im->next;
        if (im->alive && d->cur_usr && d->cur_usr->free_image) {
            d->cur_usr->free_image(d->cur_usr, im->usr_context);
        }
        im->alive = false;
        im->next = d->free_images;
        d->free_images = im;
        ring_after_free(d, im->arena_begin);
    }
    if (!d->used_images_head) {
        d->used_segs_head = d->used_segs_tail = kNullSeg;
        d->used_images_tail = nullptr;
    } else {
        d->used_segs_head = end_image->first_seg;
    }
}

bool seg_earlier(Dict *d, uint32_t dst, uint32_t src)
{
    return src == kNullSeg ||
           (dst != kNullSeg && d->segs[dst].pixels_so_far < d->segs[src].pixels_so_far);
}

/* glz_encoder_dictionary.c:593-633 */
void post_encode(Dict *d, uint32_t enc_id)
{
    uint32_t early = kNullSeg;
    for (uint32_t i = 0; i < d->max_encoders; i++) {
        if (i != enc_id && seg_earlier(d, d->encoders_heads[i], early)) {
            early = d->encoders_heads[i];
        }
    }
    if (seg_earlier(d, d->used_segs_head, early)) {
        early = d->used_segs_head;
    }
    uint32_t mine = d->encoders_heads[enc_id];
    if (seg_earlier(d, mine, early)) {
        drop_head_until(d, d->segs[early].image);
    }
    d->encoders_heads[enc_id] = kNullSeg;
}

/* ------------------------------------------------ one image, ready to launch */
struct Chunk {
    const uint8_t *ptr;
    uint32_t rows;
};

struct Prepared {
    Encoder *enc = nullptr;
    HostImage *image = nullptr;
    GlzDevJob dev;
    std::vector<uint32_t> seg_list;
    std::vector<GlzDevSegUpdate> updates;
    std::vector<GlzCopyDesc> copies; /* src -> arena, in segment order */
    uint64_t raw_bytes = 0;

    void reset()
    {
        enc = nullptr;
        image = nullptr;
        seg_list.clear();
        updates.clear();
        copies.clear();
        raw_bytes = 0;
    }
};

const char *validate(LzImageType type, int width, int height, int stride)
{
    if (type <= LZ_IMAGE_TYPE_INVALID || type > LZ_IMAGE_TYPE_RGBA) {
        return "bad image type\n"; /* glz_encoder.c:299-302 */
    }
    if (kTypeIsPlt[type]) {
        /* red_worker.c:6012 only ever hands RGB bitmaps to GLZ */
        return "palette images are not handled by the GPU GLZ path\n";
    }
    if (width <= 0 || height <= 0) {
        return "bad image size\n";
    }
    if (stride != width * kRgbBytesPerPixel[type]) {
        return "stride != width*bytes_per_pixel (rgb)\n"; /* glz_encoder.c:243-247 */
    }
    return nullptr;
}

/* glz_dictionary_pre_encode (glz_encoder_dictionary.c:550-591) plus the launch
 * description.  Called with the dictionary locked. */
const char *prepare(Encoder *enc, LzImageType type, int width, int height, int top_down,
                    int stride, const std::vector<Chunk> &chunks, GlzUsrImageContext *usr_context,
                    uint8_t *d_out, uint32_t out_cap, bool device_src, hipStream_t st, Prepared *p)
{
    Dict *d = enc->dict;
    d->cur_usr = enc->usr;

    int image_size = kTypeIsRgb[type] ? height * stride / kRgbBytesPerPixel[type]
                                      : height * stride * kPltPixelsPerByte[type];
    if ((uint32_t)image_size > d->size_limit) {
        return "image is bigger than window\n"; /* glz_encoder_dictionary.c:397-399 */
    }
    uint32_t rows_total = 0;
    for (const Chunk &c : chunks) {
        rows_total += c.rows;
    }
    if (chunks.empty() || rows_total < (uint32_t)height) {
        return "more lines failed\n"; /* glz_encoder_dictionary.c:497-501,517-520 */
    }

    HostImage *new_head = head_after_adding(d, image_size);
    if (!dict_in_use(d)) {
        drop_head_until(d, new_head);
    }

    /* arena room for every chunk, each starting on a 16-byte boundary */
    uint64_t need = 0;
    uint32_t row = 0;
    size_t nseg = 0;
    for (const Chunk &c : chunks) {
        need += align16((uint64_t)c.rows * (uint64_t)stride);
        row += c.rows;
        nseg++;
        if (row >= (uint32_t)height) {
            break;
        }
    }
    uint64_t at = 0;
    if (!ring_alloc(d, need, &at) && !(arena_regrow(d, need, st) && ring_alloc(d, need, &at))) {
        return "window arena exhausted\n";
    }

    /* glz_encoder_dictionary.c:480-548 */
    HostImage *im = take_image(d);
    im->id = d->last_image_id++;
    im->size = image_size;
    im->type = type;
    im->usr_context = usr_context;
    im->arena_begin = at;
    im->arena_end = at + need;

    const int bpp = kRgbBytesPerPixel[type];
    uint32_t sid = kNullSeg, prev = kNullSeg;
    row = 0;
    uint64_t off = at;
    for (size_t c = 0; c < nseg; c++) {
        uint64_t bytes = (uint64_t)chunks[c].rows * (uint64_t)stride;
        sid = take_seg(d);
        HostSeg &s = d->segs[sid];
        s.image = im;
        s.arena_off = off;
        s.nunits = (uint32_t)(bytes / (uint64_t)bpp);
        s.pixels_num = (uint32_t)((int64_t)image_size * chunks[c].rows / height);
        s.pixels_so_far = d->pixels_so_far;
        d->pixels_so_far += s.pixels_num;
        s.next = kNullSeg;
        if (row == 0) {
            im->first_seg = sid;
        } else {
            d->segs[prev].next = sid;
        }
        row += chunks[c].rows;
        prev = sid;
        p->seg_list.push_back(sid);
        GlzCopyDesc cd;
        cd.src = chunks[c].ptr;
        cd.dst = d->d_arena + off;
        cd.bytes = bytes;
        p->copies.push_back(cd);
        p->raw_bytes += bytes;
        off += align16(bytes);
    }
    if (d->used_segs_tail == kNullSeg) {
        d->used_segs_head = im->first_seg;
        d->used_segs_tail = sid;
    } else {
        d->segs[d->used_segs_tail].next = im->first_seg;
        d->used_segs_tail = sid;
    }
    im->alive = true;
    for (uint32_t s : p->seg_list) {
        d->dirty.push_back(s);
    }

    uint32_t head_dist;
    if (new_head) {
        d->encoders_heads[enc->id] = new_head->first_seg;
        head_dist = (uint32_t)(im->id - new_head->id);
    } else {
        d->encoders_heads[enc->id] = im->first_seg;
        head_dist = 0;
    }

    if (!ensure_dev_segs(d, st)) {
        /* the image is in the window already: it must never be referenced */
        im->alive = false;
        post_encode(d, enc->id);
        return "device segment table allocation failed\n";
    }
    for (uint32_t s : d->dirty) {
        GlzDevSegUpdate u;
        memset(&u, 0, sizeof(u));
        u.index = s;
        u.seg = dev_view(d, d->segs[s]);
        p->updates.push_back(u);
    }
    d->dirty.clear();

    p->enc = enc;
    p->image = im;
    memset(&p->dev, 0, sizeof(p->dev));
    p->dev.arena = d->d_arena;
    p->dev.htab = d->d_htab;
    p->dev.segs = d->d_segs;
    p->dev.out = d_out;
    p->dev.out_cap = out_cap;
    p->dev.image_id = im->id;
    p->dev.head_psf = d->segs[d->encoders_heads[enc->id]].pixels_so_far;
    p->dev.type = (uint32_t)type;
    p->dev.width = (uint32_t)width;
    p->dev.height = (uint32_t)height;
    p->dev.stride = (uint32_t)stride;
    p->dev.top_down = top_down ? 1u : 0u;
    p->dev.win_head_dist = head_dist;
    p->dev.nsegs = (uint32_t)p->seg_list.size();
    p->dev.nupd = (uint32_t)p->updates.size();
    /* an image of one chunk that is already in device memory is parsed where it
     * lies; its copy into the arena only serves later images */
    if (device_src && p->seg_list.size() == 1 && ((uintptr_t)chunks[0].ptr & 15) == 0) {
        p->dev.own_src = chunks[0].ptr;
        p->dev.own_seg = p->seg_list[0];
    }
    return nullptr;
}

/* An image that was added to its window but whose stream the client will never
 * receive (the launch failed, the device reported an error, the output did not
 * fit): it stays in the window as a dead image, like one removed with
 * glz_enc_dictionary_remove_image, so that no later image refers to it.  In
 * the reference this situation is fatal (usr->error is spice_critical,
 * server/red_worker.c:5578-5591). */
void kill_image(Dict *d, HostImage *im)
{
    if (im && im->alive) {
        im->alive = false;
        mark_image_dirty(d, im);
    }
}

/* A prepared job that will not be launched: its segment updates go back on the
 * dictionary's list, its image dies, its encoder lets go of the window. */
void abandon(Prepared &p)
{
    if (!p.enc) {
        return;
    }
    Dict *d = p.enc->dict;
    for (const GlzDevSegUpdate &u : p.updates) {
        d->dirty.push_back(u.index);
    }
    kill_image(d, p.image);
    post_encode(d, p.enc->id);
    p.reset();
}

/* Jobs of a batch that may run side by side: at most one per dictionary.  A
 * batch with several images for one dictionary (the display channels of one
 * client share theirs, server/red_worker.c:9677-9754) is cut into levels, the
 * k-th image of every dictionary in level k; levels run one after the other on
 * the library's stream, so each dictionary sees its images in the caller's
 * order, exactly as if they had been encoded by separate calls. */
struct Level {
    std::vector<int> idx;       /* the caller's job numbers, ascending */
    std::vector<Prepared> jobs; /* same order */
    std::vector<Prepared *> ptrs;
    Staging staging;
    size_t results_off = 0;
};

/* Batch in flight between submit and finish.  There are kFlights slots, each with its
 * own stream, side stream and staging, so that a caller can keep several batches of
 * DISJOINT dictionaries on the device: while the slowest sessions of one batch finish
 * their images, the other batches' workgroups take the places the fast ones left (a
 * parse launch lasts as long as its slowest image).  A slot belongs to the thread that
 * submitted into it until that thread's finish. */
struct Flight {
    std::vector<std::unique_ptr<Level>> levels; /* kept between batches for their buffers */
    size_t nlevels = 0;
    int njobs = 0;
    std::vector<std::unique_lock<std::recursive_mutex>> dict_locks;
    std::vector<Dict *> dicts;
    std::thread::id owner; /* the thread that submitted: its dictionary locks are its own */
    bool active = false;   /* claimed; g_fl_mutex guards owner, active, seq, njobs */
    uint64_t seq = 0;      /* order of submission */
    uint64_t t0 = 0;
    uint64_t host_ns = 0; /* host time of this batch's submit + finish, waiting excluded */
    OwnStream main, side;
    hipEvent_t fork = nullptr, join = nullptr;
};
const int kFlights = 4; /* the device runs the kernels of four hardware queues side by side */
Flight g_flights[kFlights];
std::mutex g_fl_mutex;
std::condition_variable g_fl_cv;
uint64_t g_fl_seq = 0;
uint64_t g_last_host_ns = 0;

bool flight_ready(Flight &f)
{
    /* the first two flights' four streams are ones the device runs side by side (a parse
     * that queues behind the other flight's parse, or a copy behind a parse, loses what
     * the second flight is for); further flights take what the runtime gives them */
    const int idx = (int)(&f - g_flights);
    hipStream_t pool[kSideBySide];
    if (!f.main.st && idx < 2 && side_by_side_streams(pool)) {
        f.main.st = pool[2 * idx];
        f.side.st = pool[2 * idx + 1];
    }
    return f.main.ensure() && f.side.ensure() &&
           (f.fork || check(hipEventCreate(&f.fork), "hipEventCreate")) &&
           (f.join || check(hipEventCreate(&f.join), "hipEventCreate"));
}

/* what launch() enqueues on: a stream, and for the copies beside the parse a second one
 * with the two events that order them (null: everything on the one stream) */
struct LaunchStreams {
    hipStream_t s;
    hipStream_t side;
    hipEvent_t fork, join;
    int lone_skip = 0; /* glz_launch_parse_lone's skip_blocks: differs between launches that run side by side */
};

/* Launches of at most this many images use the kernel for an under-filled GPU: the number of
 * CUs of the device unless spice_gpu_glz_lone_max_jobs or SPICE_GPU_GLZ_LONE_MAX says otherwise
 * (0: never). */
std::atomic<int> g_lone_max{-1};

int lone_max_jobs()
{
    int v = g_lone_max.load();
    if (v < 0) {
        v = 256;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, current_device() < 0 ? 0 : current_device()) == hipSuccess &&
            prop.multiProcessorCount > 0) {
            v = prop.multiProcessorCount;
        }
        const char *e = getenv("SPICE_GPU_GLZ_LONE_MAX");
        if (e && *e) {
            v = atoi(e);
        }
        g_lone_max.store(v);
    }
    return v;
}

/* Lays jobs, pools and copy descriptors out in the staging buffer, uploads
 * them and enqueues copy + parse.  copy_on_device: the chunk sources are
 * device pointers (batch path); otherwise the caller has already enqueued
 * host-to-arena copies on the stream. */
bool launch(Prepared *const *jobs, size_t n, Staging &st, size_t *results_off, bool copy_on_device,
            const LaunchStreams &ls)
{
    hipStream_t s = ls.s;
    size_t off_jobs = 0;
    size_t off_u32 = off_jobs + sizeof(GlzDevJob) * n;
    size_t n_u32 = 0, n_upd = 0, n_copy = 0;
    for (size_t j = 0; j < n; j++) {
        const Prepared &p = *jobs[j];
        n_u32 += p.seg_list.size();
        n_upd += p.updates.size();
        n_copy += p.copies.size();
    }
    size_t off_upd = (off_u32 + sizeof(uint32_t) * n_u32 + 15) & ~(size_t)15;
    size_t off_copy = off_upd + sizeof(GlzDevSegUpdate) * n_upd;
    size_t off_res = off_copy + sizeof(GlzCopyDesc) * n_copy;
    size_t total = off_res + sizeof(int32_t) * n;
    if (!st.reserve(total)) {
        return false;
    }
    uint32_t *u32 = (uint32_t *)(st.host + off_u32);
    GlzDevSegUpdate *upd = (GlzDevSegUpdate *)(st.host + off_upd);
    GlzCopyDesc *cp = (GlzCopyDesc *)(st.host + off_copy);
    GlzDevJob *dj = (GlzDevJob *)(st.host + off_jobs);
    size_t iu = 0, iup = 0, ic = 0;
    uint64_t max_copy = 0;
    /* A workgroup parses one image and the hardware starts workgroups in index
     * order as slots free up, so a batch larger than the GPU holds at once
     * finishes soonest with the expensive images first.  The expected cost of an
     * image is its size times what its session's previous image compressed to
     * (stops per pixel follow the compressed size).  Results keep the callers'
     * job order. */
    std::vector<uint32_t> order(n);
    std::vector<double> cost(n);
    for (size_t j = 0; j < n; j++) {
        order[j] = (uint32_t)j;
        cost[j] = (double)jobs[j]->raw_bytes * (jobs[j]->enc ? jobs[j]->enc->dict->cost_hint : 1.0);
    }
    std::stable_sort(order.begin(), order.end(),
                     [&cost](uint32_t a, uint32_t b) { return cost[a] > cost[b]; });
    for (size_t pos = 0; pos < n; pos++) {
        const size_t j = order[pos];
        Prepared &p = *jobs[j];
        p.dev.seg_list_off = (uint32_t)iu;
        p.dev.upd_off = (uint32_t)iup;
        p.dev.result = (int32_t *)(st.dev + off_res) + j;
        for (uint32_t s : p.seg_list) u32[iu++] = s;
        for (const GlzDevSegUpdate &u : p.updates) upd[iup++] = u;
        for (const GlzCopyDesc &c : p.copies) {
            cp[ic++] = c;
            max_copy = c.bytes > max_copy ? c.bytes : max_copy;
        }
        dj[pos] = p.dev;
    }
    if (!check(hipMemcpyAsync(st.dev, st.host, off_res, hipMemcpyHostToDevice, s), "upload jobs")) {
        return false;
    }
    /* When every image of the batch is parsed from its caller's buffer, the copies into the
     * arenas only serve later images: they follow the parse, on the side stream beside it
     * (stream() waits for them at the end, so whoever synchronises it has both) or, without
     * one, behind it on the same stream, where they run beside the other flights' parses. */
    bool own_all = copy_on_device && n_copy != 0;
    for (size_t j = 0; j < n; j++) {
        own_all = own_all && jobs[j]->dev.own_src != nullptr;
    }
    const bool beside = own_all && ls.side != nullptr;
    uint64_t per_block = 256ull * 16 * 8; /* 8 vectors per lane */
    unsigned blocks = (unsigned)((max_copy + per_block - 1) / per_block);
    blocks = blocks < 1 ? 1 : (blocks > 1024 ? 1024 : blocks);
    if (copy_on_device && n_copy && !own_all) {
        glz_launch_copy((const GlzCopyDesc *)(st.dev + off_copy), (int)n_copy, blocks, s);
    }
    if (beside && !check(hipEventRecord(ls.fork, s), "event")) {
        return false;
    }
    unsigned type_mask = 0;
    for (size_t j = 0; j < n; j++) {
        type_mask |= 1u << jobs[j]->dev.type;
    }
    /* Fewer images than the GPU has CUs: every image gets a CU to itself and the kernel that
     * runs scan, walk and emit side by side (glz_lone.hip); a full GPU is served best by
     * the kernel that fits eight images on a CU.  Same streams either way. */
    if ((int)n <= lone_max_jobs()) {
        glz_launch_parse_lone((const GlzDevJob *)(st.dev + off_jobs), (int)n, type_mask,
                              (const uint32_t *)(st.dev + off_u32),
                              (const GlzDevSegUpdate *)(st.dev + off_upd), s, ls.lone_skip);
    } else {
        glz_launch_parse((const GlzDevJob *)(st.dev + off_jobs), (int)n, type_mask,
                         (const uint32_t *)(st.dev + off_u32),
                         (const GlzDevSegUpdate *)(st.dev + off_upd), s);
    }
    if (beside) {
        /* after the parse, so that the parse's workgroups take their places first
         * and the copy fills what is left of the CUs */
        hipStream_t cs = ls.side;
        if (!check(hipStreamWaitEvent(cs, ls.fork, 0), "wait event")) {
            return false;
        }
        glz_launch_copy((const GlzCopyDesc *)(st.dev + off_copy), (int)n_copy, blocks, cs);
        if (!check(hipEventRecord(ls.join, cs), "event")) {
            return false;
        }
    } else if (own_all) {
        glz_launch_copy((const GlzCopyDesc *)(st.dev + off_copy), (int)n_copy, blocks, s);
    }
    if (!check(hipGetLastError(), "glz launch")) {
        return false;
    }
    if (!check(hipMemcpyAsync(st.host + off_res, st.dev + off_res, sizeof(int32_t) * n,
                              hipMemcpyDeviceToHost, s), "download results")) {
        return false;
    }
    if (beside && !check(hipStreamWaitEvent(s, ls.join, 0), "wait event")) {
        return false;
    }
    *results_off = off_res;
    return true;
}

/* Worker threads that call glz_encode at about the same time (one RedWorker
 * thread per display, server/red_dispatcher.c:1148) are served by one parse
 * launch: every caller has prepared its image and uploaded its rows, queues the
 * job, and the first one in becomes a round's leader, launches all queued jobs
 * together and hands each caller its result.  A device runs the kernels of few
 * streams side by side (it has a handful of hardware queues), so separate
 * launches from sixteen threads would mostly wait for one another; one launch
 * has a workgroup per caller.  Up to kRounds rounds run at a time, each on a
 * stream of its own: a caller that arrives while a round runs does not wait for
 * that round's images -- a thread with small images is not held up by its
 * neighbours' large ones unless they arrive within the same half millisecond --
 * and callers beyond that form the next round to start.  Every caller holds
 * its own dictionary's lock meanwhile, so the jobs of all rounds are on
 * distinct dictionaries. */
struct Waiting {
    Prepared *job;
    int32_t size = 0;
    bool ok = false;
    bool done = false;
    bool taken = false; /* a leader has this job in its round */
};

/* the device runs kSideBySide streams side by side: all but one carry rounds, the last the
 * callers' row uploads and stream downloads, which must not queue behind a round's kernel */
const int kRounds = kSideBySide - 1;
struct Combiner {
    std::mutex m;
    std::condition_variable cv;
    std::vector<Waiting *> queue;
    bool gathering = false; /* a leader is collecting its round: arrivals join it */
    bool slot_busy[kRounds] = {false, false, false};
    hipStream_t slot_stream[kSideBySide]; /* the rounds' streams: on different hardware queues */
    bool streams_made = false, streams_ok = false;
    int entering = 0; /* callers that will queue a job shortly */
};
Combiner g_combiner;

/* the stream a synchronous caller's copies go on (its encoder's own when the probe failed) */
hipStream_t combiner_copy_stream(Encoder *enc)
{
    Combiner &c = g_combiner;
    std::lock_guard<std::mutex> lk(c.m);
    if (!c.streams_made) {
        c.streams_made = true;
        c.streams_ok = side_by_side_streams(c.slot_stream);
    }
    return c.streams_ok ? c.slot_stream[kSideBySide - 1] : enc->st;
}

void combiner_enter()
{
    std::lock_guard<std::mutex> lk(g_combiner.m);
    g_combiner.entering++;
}

void combiner_leave()
{
    std::lock_guard<std::mutex> lk(g_combiner.m);
    g_combiner.entering--;
    g_combiner.cv.notify_all();
}

/* Called by a thread that has counted itself in with combiner_enter().  Returns
 * true with *size set when the job's stream is in its encoder's device buffer. */
bool combiner_run(Encoder *enc, Prepared *job, int32_t *size)
{
    Combiner &c = g_combiner;
    Waiting me;
    me.job = job;
    std::unique_lock<std::mutex> lk(c.m);
    c.entering--;
    c.queue.push_back(&me);
    c.cv.notify_all();
    while (!me.done) {
        int slot = -1;
        if (!me.taken && !c.gathering) {
            for (int k = 0; k < kRounds && slot < 0; k++) {
                slot = c.slot_busy[k] ? -1 : k;
            }
        }
        if (slot < 0) {
            c.cv.wait(lk);
            continue;
        }
        /* lead a round: give the callers whose rows are already on their way to the
         * device a moment (a caller counts itself in only after it has enqueued its upload,
         * so what is waited for is the rest of a copy, never another caller's allocations
         * or window bookkeeping) */
        c.slot_busy[slot] = true;
        c.gathering = true;
        const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(500);
        while (c.entering > 0 && c.cv.wait_until(lk, until) != std::cv_status::timeout) {
        }
        std::vector<Waiting *> round;
        round.swap(c.queue);
        for (Waiting *w : round) {
            w->taken = true;
        }
        c.gathering = false;
        lk.unlock();
        std::vector<Prepared *> ptrs;
        for (Waiting *w : round) {
            ptrs.push_back(w->job);
        }
        size_t res_off = 0;
        /* this round's alone until slot_busy is cleared (the callers made the streams) */
        const hipStream_t rs = c.streams_ok ? c.slot_stream[slot] : enc->st;
        bool ok =
                  launch(ptrs.data(), ptrs.size(), enc->staging, &res_off, false,
                         LaunchStreams{rs, nullptr, nullptr, nullptr, 2 * slot});
        ok = check(hipStreamSynchronize(rs), "glz sync") && ok;
        const int32_t *res = (const int32_t *)(enc->staging.host + res_off);
        lk.lock();
        for (size_t k = 0; k < round.size(); k++) {
            round[k]->ok = ok;
            round[k]->size = ok ? res[k] : 0;
            round[k]->done = true;
        }
        c.slot_busy[slot] = false;
        c.cv.notify_all();
    }
    *size = me.size;
    return me.ok;
}

void usr_error(GlzEncoderUsrContext *usr, const char *msg)
{
    set_error("glz: %s", msg);
    if (usr && usr->error) {
        usr->error(usr, "%s", msg);
    }
}

} // namespace

/* ====================================================================== C ABI */
extern "C" {

GlzEncDictContext *glz_enc_dictionary_create(uint32_t size, uint32_t max_encoders,
                                             GlzEncoderUsrContext *usr)
{
    if (size > kMaxWindow || max_encoders == 0 || !ensure_init()) {
        if (size > kMaxWindow) {
            set_error("glz: window of %u pixels exceeds the 2^25 cap", size);
        }
        return nullptr;
    }
    Dict *d = new Dict();
    d->cur_usr = usr;
    d->size_limit = size;
    d->max_encoders = max_encoders;
    d->segs.resize(kInitSegs);
    d->encoders_heads.assign(max_encoders, kNullSeg);
    /* FIFO ring: the window in its widest pixel format; re-made larger by
     * arena_regrow the first time the ring's wrap point wastes too much */
    d->arena_cap = (uint64_t)size * 4ull + arena_slack(size);
    bool ok = check(hipMalloc((void **)&d->d_arena, d->arena_cap + 64), "hipMalloc arena") &&
              check(hipMalloc((void **)&d->d_htab, sizeof(uint2) * (size_t)GLZ_HASH_SIZE), "hipMalloc htab");
    ok = ok && ensure_dev_segs(d, stream());
    if (!ok || !dict_reset(d)) {
        dict_free_device(d);
        delete d;
        return nullptr;
    }
    return (GlzEncDictContext *)d;
}

void glz_enc_dictionary_destroy(GlzEncDictContext *opaque_dict, GlzEncoderUsrContext *usr)
{
    Dict *d = (Dict *)opaque_dict;
    if (!d) {
        return;
    }
    d->cur_usr = usr;
    release_all_images(d);
    while (d->free_images) {
        HostImage *im = d->free_images;
        d->free_images = im->next;
        delete im;
    }
    dict_free_device(d);
    delete d;
}

uint32_t glz_enc_dictionary_get_size(GlzEncDictContext *opaque_dict)
{
    return opaque_dict ? ((Dict *)opaque_dict)->size_limit : 0;
}

void glz_enc_dictionary_get_restore_data(GlzEncDictContext *opaque_dict,
                                         GlzEncDictRestoreData *out_data,
                                         GlzEncoderUsrContext *usr)
{
    Dict *d = (Dict *)opaque_dict;
    d->cur_usr = usr;
    out_data->last_image_id = d->last_image_id;
    out_data->max_encoders = d->max_encoders;
    out_data->size = d->size_limit;
}

GlzEncDictContext *glz_enc_dictionary_restore(GlzEncDictRestoreData *restore_data,
                                              GlzEncoderUsrContext *usr)
{
    if (!restore_data) {
        return nullptr;
    }
    Dict *d = (Dict *)glz_enc_dictionary_create(restore_data->size, restore_data->max_encoders, usr);
    if (d) {
        d->last_image_id = restore_data->last_image_id;
    }
    return (GlzEncDictContext *)d;
}

void glz_enc_dictionary_reset(GlzEncDictContext *opaque_dict, GlzEncoderUsrContext *usr)
{
    Dict *d = (Dict *)opaque_dict;
    std::lock_guard<std::recursive_mutex> lock(d->lock);
    d->cur_usr = usr;
    dict_reset(d);
}

void glz_enc_dictionary_remove_image(GlzEncDictContext *opaque_dict,
                                     GlzEncDictImageContext *opaque_image,
                                     GlzEncoderUsrContext *usr)
{
    Dict *d = (Dict *)opaque_dict;
    HostImage *im = (HostImage *)opaque_image;
    std::lock_guard<std::recursive_mutex> lock(d->lock);
    d->cur_usr = usr;
    if (im->alive) {
        im->alive = false; /* glz_encoder_dictionary.c:142-145 */
        mark_image_dirty(d, im);
    }
}

void spice_gpu_glz_dictionary_info(GlzEncDictContext *opaque_dict, SpiceGpuGlzDictInfo *out)
{
    Dict *d = (Dict *)opaque_dict;
    std::lock_guard<std::recursive_mutex> lock(d->lock);
    memset(out, 0, sizeof(*out));
    out->arena_bytes = d->arena_cap;
    out->table_bytes = sizeof(uint2) * (uint64_t)GLZ_HASH_SIZE + sizeof(GlzDevSeg) * (uint64_t)d->d_segs_cap;
    out->segments = (uint32_t)d->segs.size();
    out->arena_regrows = d->arena_regrows;
    for (HostImage *im = d->used_images_head; im; im = im->next) {
        out->images++;
        out->window_pixels += (uint32_t)im->size;
    }
}

size_t spice_gpu_glz_dictionary_dump(GlzEncDictContext *opaque_dict, uint64_t *out, size_t cap)
{
    Dict *d = (Dict *)opaque_dict;
    std::lock_guard<std::recursive_mutex> lock(d->lock);
    size_t n = 0;
    auto put = [&](uint64_t v) {
        if (n < cap) {
            out[n] = v;
        }
        n++;
    };
    put(d->size_limit);
    put(d->max_encoders);
    put(d->last_image_id);
    put(d->pixels_so_far);
    put(d->segs.size());
    put(d->used_segs_head);
    put(d->used_segs_tail);
    for (uint32_t h : d->encoders_heads) {
        put(h);
    }
    uint64_t nimages = 0;
    for (HostImage *im = d->used_images_head; im; im = im->next) {
        nimages++;
    }
    put(nimages);
    for (HostImage *im = d->used_images_head; im; im = im->next) {
        put(im->id);
        put((uint64_t)im->size);
        put(im->alive ? 1 : 0);
        put((uint64_t)(uintptr_t)im->usr_context);
        put((uint64_t)im->type);
        uint64_t nsegs = 0;
        for (uint32_t sid = im->first_seg; sid != kNullSeg && d->segs[sid].image == im;
             sid = d->segs[sid].next) {
            nsegs++;
        }
        put(nsegs);
        for (uint32_t sid = im->first_seg; sid != kNullSeg && d->segs[sid].image == im;
             sid = d->segs[sid].next) {
            put(sid);
            put(d->segs[sid].pixels_so_far);
            put(d->segs[sid].pixels_num);
            put(d->segs[sid].next);
        }
    }
    uint64_t nfree = 0;
    for (uint32_t sid = d->free_segs_head; sid != kNullSeg; sid = d->segs[sid].next) {
        nfree++;
    }
    put(nfree);
    for (uint32_t sid = d->free_segs_head; sid != kNullSeg; sid = d->segs[sid].next) {
        put(sid);
    }
    return n;
}

GlzEncoderContext *glz_encoder_create(uint8_t id, GlzEncDictContext *dictionary,
                                      GlzEncoderUsrContext *usr)
{
    /* glz_encoder.c:131-134 */
    if (!usr || !usr->error || !usr->warn || !usr->info || !usr->malloc || !usr->free ||
        !usr->more_space) {
        return nullptr;
    }
    Dict *d = (Dict *)dictionary;
    if (!d || id >= d->max_encoders) {
        return nullptr;
    }
    if (!ensure_init()) {
        return nullptr;
    }
    Encoder *e = new Encoder();
    e->usr = usr;
    e->id = id;
    e->dict = d;
    if (!check(hipStreamCreateWithFlags(&e->st, hipStreamNonBlocking), "hipStreamCreate")) {
        delete e;
        return nullptr;
    }
    return (GlzEncoderContext *)e;
}

void glz_encoder_destroy(GlzEncoderContext *opaque_encoder)
{
    Encoder *e = (Encoder *)opaque_encoder;
    if (!e) {
        return;
    }
    if (e->st) {
        hipStreamSynchronize(e->st);
        hipStreamDestroy(e->st);
    }
    e->staging.release();
    e->out.release();
    e->bounce.release();
    delete e;
}

size_t spice_gpu_glz_out_bound(LzImageType type, int width, int height)
{
    /* worst case per pixel: a count byte plus the longest reference (7 bytes)
     * for every 2 pixels, or a literal; the alpha plane adds one more pass */
    size_t px = (size_t)width * (size_t)height;
    size_t per = type == LZ_IMAGE_TYPE_RGBA ? 9 : 5;
    return GLZ_HEADER_BYTES + px * per + 64;
}

int glz_encode(GlzEncoderContext *opaque_encoder, LzImageType type, int width, int height,
               int top_down, uint8_t *lines, unsigned int num_lines, int stride,
               uint8_t *io_ptr, unsigned int num_io_bytes, GlzUsrImageContext *usr_context,
               GlzEncDictImageContext **o_enc_dict_context)
{
    Encoder *enc = (Encoder *)opaque_encoder;
    Dict *d = enc->dict;
    uint64_t t0 = now_ns();
    if (!ensure_init()) {
        usr_error(enc->usr, "no usable GPU\n");
        return 0;
    }
    const char *bad = validate(type, width, height, stride);
    if (bad) {
        usr_error(enc->usr, bad);
        return 0;
    }

    /* glz_encoder_dictionary.c:497-525: first chunk as passed, the rest pulled */
    std::vector<Chunk> chunks;
    {
        uint32_t row = 0;
        uint8_t *cur = lines;
        int n = (int)num_lines;
        while (row < (uint32_t)height) {
            if (n <= 0) {
                n = enc->usr->more_lines ? enc->usr->more_lines(enc->usr, &cur) : 0;
                if (n <= 0) {
                    usr_error(enc->usr, "more lines failed\n");
                    return 0;
                }
            }
            chunks.push_back(Chunk{cur, (uint32_t)n});
            row += (uint32_t)n;
            n = 0;
        }
    }

    /* the dictionary's lock is all this call needs: it works on the encoder's own
     * stream and buffers, so encoders of different dictionaries do not wait for
     * each other (a batch in flight holds the locks of its dictionaries) */
    std::lock_guard<std::recursive_mutex> dict_lock(d->lock);
    if (d->in_flight) {
        /* only the thread that holds the lock recursively gets here: its own batch */
        usr_error(enc->usr, "this thread has a batch in flight on the dictionary\n");
        return 0;
    }

    size_t bound = spice_gpu_glz_out_bound(type, width, height);
    if (!enc->out.reserve(bound) || !enc->bounce.reserve(bound)) {
        usr_error(enc->usr, "device output buffer allocation failed\n");
        return 0;
    }
    std::vector<Prepared> jobs(1);
    const char *err = prepare(enc, type, width, height, top_down, stride, chunks, usr_context,
                              enc->out.ptr, (uint32_t)bound, false, enc->st, &jobs[0]);
    if (err) {
        usr_error(enc->usr, err);
        return 0;
    }
    if (o_enc_dict_context) {
        *o_enc_dict_context = (GlzEncDictImageContext *)jobs[0].image;
    }

    hipStream_t s = combiner_copy_stream(enc);
    bool ok = true;
    for (const GlzCopyDesc &c : jobs[0].copies) {
        ok = ok && check(hipMemcpyAsync(c.dst, c.src, c.bytes, hipMemcpyHostToDevice, s), "upload rows");
    }
    /* from here until the job is queued, a round's leader waits for this caller (briefly) */
    struct Entering {
        bool counted = true;
        Entering() { combiner_enter(); }
        ~Entering()
        {
            if (counted) {
                combiner_leave();
            }
        }
    } entering;
    ok = ok && check(hipStreamSynchronize(s), "upload rows");
    int32_t size = 0;
    if (ok) {
        entering.counted = false; /* combiner_run counts this caller out itself */
        ok = combiner_run(enc, &jobs[0], &size);
    }
    if (ok && size > 0) {
        ok = check(hipMemcpyAsync(enc->bounce.ptr, enc->out.ptr, (size_t)size,
                                  hipMemcpyDeviceToHost, s), "download stream") &&
             check(hipStreamSynchronize(s), "glz sync");
    }
    if (!ok) {
        hipStreamSynchronize(s); /* nothing of this image may still be running */
        /* the segment updates prepare() took off the dictionary's list may not have reached
         * the device: they go back on it, or the device's table would keep evicted segments
         * alive */
        for (const GlzDevSegUpdate &u : jobs[0].updates) {
            d->dirty.push_back(u.index);
        }
    }
    if (!ok || size <= 0) {
        kill_image(d, jobs[0].image);
        if (o_enc_dict_context) {
            *o_enc_dict_context = nullptr;
        }
    }
    post_encode(d, enc->id);
    if (!ok || size <= 0) {
        usr_error(enc->usr, ok ? "no more bytes\n" : "device failure\n");
        return 0;
    }

    /* hand the stream out through io_ptr and then more_space (glz_encoder.c:52-73) */
    {
        size_t left = (size_t)size;
        const uint8_t *src = enc->bounce.ptr;
        uint8_t *dst = io_ptr;
        size_t room = num_io_bytes;
        while (left) {
            if (room == 0) {
                int got = enc->usr->more_space(enc->usr, &dst);
                if (got <= 0) {
                    usr_error(enc->usr, "no more bytes\n");
                    return 0;
                }
                room = (size_t)got;
            }
            size_t n = left < room ? left : room;
            memcpy(dst, src, n);
            src += n;
            dst += n;
            left -= n;
            room -= n;
        }
    }
    stat_add(SPICE_GPU_STAT_GLZ, jobs[0].raw_bytes, (uint64_t)size, now_ns() - t0);
    return size;
}

int spice_gpu_glz_submit(SpiceGpuGlzJob *jobs, int njobs)
{
    if (!ensure_init()) {
        return -1;
    }
    if (njobs <= 0) {
        set_error("glz batch: no jobs");
        return -1;
    }
    const uint64_t t_in = now_ns();
    /* everything that can be checked without touching a window is checked first:
     * a bad job must not leave other jobs' images half added */
    std::vector<Dict *> dicts((size_t)njobs);
    for (int j = 0; j < njobs; j++) {
        SpiceGpuGlzJob &job = jobs[j];
        Encoder *enc = (Encoder *)job.encoder;
        job.out_size = -1;
        job.dict_image = nullptr;
        if (!enc) {
            set_error("glz batch job %d: null encoder", j);
            return -1;
        }
        const char *err = validate(job.type, job.width, job.height, job.stride);
        if (!err && (job.nchunks <= 0 || !job.chunk_ptrs || !job.chunk_rows || !job.out)) {
            err = "no chunks or no output buffer\n";
        }
        if (!err) {
            uint64_t rows = 0;
            for (int c = 0; c < job.nchunks; c++) {
                rows += job.chunk_rows[c];
                if (!job.chunk_ptrs[c]) {
                    err = "null chunk\n";
                }
            }
            const int64_t px = (int64_t)job.height * job.stride / kRgbBytesPerPixel[job.type];
            if (rows < (uint64_t)job.height) {
                err = "more lines failed\n"; /* glz_encoder_dictionary.c:497-501,517-520 */
            } else if (px > (int64_t)enc->dict->size_limit) {
                err = "image is bigger than window\n"; /* glz_encoder_dictionary.c:397-399 */
            }
        }
        if (err) {
            set_error("glz batch job %d: %s", j, err);
            return -1;
        }
        dicts[(size_t)j] = enc->dict;
    }
    /* the distinct dictionaries, in address order (two batches that name the same
     * dictionaries in different orders must not wait for each other) */
    std::vector<Dict *> distinct(dicts);
    std::sort(distinct.begin(), distinct.end());
    distinct.erase(std::unique(distinct.begin(), distinct.end()), distinct.end());
    /* level of a job = how many earlier jobs of the batch use its dictionary */
    std::vector<uint32_t> seen(distinct.size(), 0);
    std::vector<uint32_t> level_of((size_t)njobs);
    size_t nlevels = 0;
    for (int j = 0; j < njobs; j++) {
        const size_t k = (size_t)(std::lower_bound(distinct.begin(), distinct.end(), dicts[(size_t)j]) -
                                  distinct.begin());
        level_of[(size_t)j] = seen[k]++;
        nlevels = seen[k] > nlevels ? seen[k] : nlevels;
    }
    /* lock order everywhere: dictionaries, then a flight slot */
    std::vector<std::unique_lock<std::recursive_mutex>> dict_locks;
    dict_locks.reserve(distinct.size());
    for (Dict *d : distinct) {
        dict_locks.emplace_back(d->lock);
    }
    for (Dict *d : distinct) {
        if (d->in_flight) {
            /* the locks are recursive: only this thread's own earlier batch gets here */
            set_error("glz batch: a dictionary of this batch is in another batch in flight "
                      "(two batches in flight must use disjoint dictionaries)");
            return -1;
        }
    }
    /* a free flight slot; a thread may hold both, a third submit is refused, and while
     * other threads hold them this one waits for a finish */
    Flight *fp = nullptr;
    {
        std::unique_lock<std::mutex> lk(g_fl_mutex);
        const std::thread::id me = std::this_thread::get_id();
        for (;;) {
            int mine = 0;
            fp = nullptr;
            for (Flight &c : g_flights) {
                if (!c.active) {
                    fp = fp ? fp : &c;
                } else if (c.owner == me) {
                    mine++;
                }
            }
            if (mine >= kFlights) {
                set_error("glz batch: this thread has %d batches in flight already (finish one first)", mine);
                return -1;
            }
            if (fp) {
                break;
            }
            g_fl_cv.wait(lk);
        }
        fp->active = true;
        fp->owner = me;
        fp->seq = ++g_fl_seq;
        fp->njobs = njobs;
    }
    Flight &f = *fp;
    auto give_back = [&f]() {
        std::lock_guard<std::mutex> lk(g_fl_mutex);
        f.active = false;
        f.nlevels = 0;
        g_fl_cv.notify_all();
    };
    if (!flight_ready(f)) {
        give_back();
        return -1;
    }
    const LaunchStreams ls{f.main.st, f.side.st, f.fork, f.join};
    while (f.levels.size() < nlevels) {
        f.levels.emplace_back(new Level());
    }
    for (size_t l = 0; l < nlevels; l++) {
        f.levels[l]->idx.clear();
    }
    for (int j = 0; j < njobs; j++) {
        f.levels[level_of[(size_t)j]]->idx.push_back(j);
    }
    f.nlevels = nlevels;
    f.t0 = t_in;

    bool ok = true;
    std::vector<Chunk> chunks;
    for (size_t l = 0; l < nlevels && ok; l++) {
        Level &lv = *f.levels[l];
        if (lv.jobs.size() < lv.idx.size()) {
            lv.jobs.resize(lv.idx.size());
        }
        size_t done = 0;
        for (; done < lv.idx.size() && ok; done++) {
            const int j = lv.idx[done];
            SpiceGpuGlzJob &job = jobs[j];
            Prepared &p = lv.jobs[done];
            p.reset();
            chunks.clear();
            for (int c = 0; c < job.nchunks; c++) {
                chunks.push_back(Chunk{(const uint8_t *)job.chunk_ptrs[c], job.chunk_rows[c]});
            }
            const char *err = prepare((Encoder *)job.encoder, job.type, job.width, job.height,
                                      job.top_down, job.stride, chunks, job.usr_context,
                                      (uint8_t *)job.out, job.out_capacity, true, ls.s, &p);
            if (err) {
                set_error("glz batch job %d: %s", j, err);
                p.reset();
                ok = false;
                break;
            }
            job.dict_image = (GlzEncDictImageContext *)p.image;
        }
        if (ok) {
            lv.ptrs.resize(lv.idx.size());
            for (size_t k = 0; k < lv.idx.size(); k++) {
                lv.ptrs[k] = &lv.jobs[k];
            }
            ok = launch(lv.ptrs.data(), lv.idx.size(), lv.staging, &lv.results_off, true, ls);
            done = lv.idx.size();
        }
        if (ok) {
            /* glz_dictionary_post_encode: the window bookkeeping does not depend on
             * what the device produces, and the next level's images must see it */
            for (size_t k = 0; k < lv.idx.size(); k++) {
                post_encode(lv.jobs[k].enc->dict, lv.jobs[k].enc->id);
            }
        } else {
            /* nothing of this level runs (a failed launch is waited for below) */
            for (size_t k = 0; k < done && k < lv.jobs.size(); k++) {
                abandon(lv.jobs[k]);
            }
            /* images of the levels already enqueued: their streams are never
             * handed out, so they die too */
            hipStreamSynchronize(ls.s);
            for (size_t e = 0; e < l; e++) {
                Level &el = *f.levels[e];
                for (size_t k = 0; k < el.idx.size(); k++) {
                    kill_image(el.jobs[k].enc->dict, el.jobs[k].image);
                }
            }
        }
    }
    if (!ok) {
        for (int j = 0; j < njobs; j++) {
            jobs[j].dict_image = nullptr;
        }
        give_back();
        return -1;
    }
    for (Dict *d : distinct) {
        d->in_flight = true;
    }
    f.dicts = distinct;
    f.dict_locks = std::move(dict_locks);
    f.host_ns = now_ns() - t_in;
    return 0;
}

/* Completes the calling thread's OLDEST batch in flight: jobs / njobs are that batch's. */
int spice_gpu_glz_finish(SpiceGpuGlzJob *jobs, int njobs)
{
    Flight *fp = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_fl_mutex);
        const std::thread::id me = std::this_thread::get_id();
        bool others = false;
        for (Flight &c : g_flights) {
            if (c.active && c.owner == me) {
                fp = (!fp || c.seq < fp->seq) ? &c : fp;
            } else if (c.active) {
                others = true;
            }
        }
        if (!fp) {
            /* a batch holds its dictionaries' locks, and a lock is released by the thread
             * that took it; another thread's batch stays in flight for its own thread */
            set_error(others ? "glz batch: finish must be called by the thread that submitted the batch"
                             : "glz batch: finish without a matching submit");
            return -1;
        }
        if (njobs != fp->njobs) {
            set_error("glz batch: finish does not match this thread's oldest batch in flight (%d jobs, not %d)",
                      fp->njobs, njobs);
            return -1;
        }
    }
    Flight &f = *fp;
    bool ok = check(hipStreamSynchronize(f.main.st), "glz batch sync");
    const uint64_t t_sync = now_ns();
    int rc = ok ? 0 : -1;
    uint64_t dt = t_sync - f.t0;
    for (size_t l = 0; l < f.nlevels; l++) {
        Level &lv = *f.levels[l];
        const int32_t *res = (const int32_t *)(lv.staging.host + lv.results_off);
        for (size_t k = 0; k < lv.idx.size(); k++) {
            const int j = lv.idx[k];
            Prepared &p = lv.jobs[k];
            Dict *d = p.enc->dict;
            jobs[j].out_size = ok ? res[k] : -1;
            if (jobs[j].out_size <= 0) {
                if (ok) {
                    set_error("glz batch job %d: device error %d", j, jobs[j].out_size);
                }
                kill_image(d, p.image);
                jobs[j].dict_image = nullptr;
                rc = -1;
            } else {
                stat_add(SPICE_GPU_STAT_GLZ, p.raw_bytes, (uint64_t)jobs[j].out_size,
                         dt / (uint64_t)njobs);
                d->cost_hint = (double)jobs[j].out_size / (double)p.raw_bytes;
            }
        }
    }
    for (Dict *d : f.dicts) {
        d->in_flight = false;
    }
    f.dicts.clear();
    f.dict_locks.clear();
    f.host_ns += now_ns() - t_sync;
    {
        std::lock_guard<std::mutex> lk(g_fl_mutex);
        g_last_host_ns = f.host_ns;
        f.nlevels = 0;
        f.active = false;
        g_fl_cv.notify_all();
    }
    return rc;
}

void spice_gpu_glz_lone_max_jobs(int njobs)
{
    g_lone_max.store(njobs);
}

uint64_t spice_gpu_glz_last_batch_host_ns(void)
{
    std::lock_guard<std::mutex> lk(g_fl_mutex);
    return g_last_host_ns;
}

int spice_gpu_glz_encode_batch(SpiceGpuGlzJob *jobs, int njobs)
{
    if (spice_gpu_glz_submit(jobs, njobs) != 0) {
        return -1;
    }
    return spice_gpu_glz_finish(jobs, njobs);
}

} // extern "C"
