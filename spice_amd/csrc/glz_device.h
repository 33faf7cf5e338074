/*
 * Layout of the GLZ state shared between the host bookkeeping (glz_host.cpp)
 * and the parse kernel (glz_kernels.hip).  Everything the kernel reads lives
 * in HBM:
 *
 *   arena   the window's pixel copies; one contiguous, 16-byte aligned run per
 *           image chunk (= window segment)
 *   htab    2^20 entries {segment index, pixel index}, the reference's HashEntry
 *           (server/glz_encoder_dictionary_protected.h:34-39,74-77)
 *   segs    one GlzDevSeg per segment slot, the device view of WindowImageSegment
 *           + the WindowImage fields reference validation needs
 *           (server/glz_encoder_dictionary_protected.h:45-71,176-184)
 */
#ifndef SPICE_GLZ_DEVICE_H
#define SPICE_GLZ_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#define GLZ_HASH_LOG 20
#define GLZ_HASH_SIZE (1u << GLZ_HASH_LOG)
#define GLZ_HASH_MASK (GLZ_HASH_SIZE - 1)
#define GLZ_MAX_COPY 32
#define GLZ_HEADER_BYTES 33

#define GLZ_SEG_TYPE_MASK 0x0fu
#define GLZ_SEG_ALIVE 0x10u
#define GLZ_SEG_USED 0x20u

struct GlzDevSeg {
    uint64_t pixels_so_far;   /* pixels that entered the window before this segment */
    uint64_t image_id;
    uint32_t arena_off16;     /* byte offset of the segment in the arena, / 16 */
    uint32_t nunits;          /* pixels in the segment */
    uint32_t img_first_delta; /* pixels_so_far minus that of the image's first segment */
    uint32_t flags;           /* LzImageType | GLZ_SEG_ALIVE | GLZ_SEG_USED */
};

struct GlzDevSegUpdate {
    uint32_t index;
    uint32_t pad;
    GlzDevSeg seg;
};

/* One image to parse.  Jobs, their segment lists and their table updates are
 * staged in one buffer that is uploaded with a single copy. */
struct GlzDevJob {
    uint8_t *arena;
    uint2 *htab;
    GlzDevSeg *segs;
    uint8_t *out;
    int32_t *result;          /* compressed size, or a negative error */
    uint64_t image_id;
    uint64_t head_psf;        /* pixels_so_far of this encoder's window head */
    uint32_t out_cap;
    uint32_t type, width, height, stride, top_down;
    uint32_t win_head_dist;
    uint32_t nsegs, seg_list_off;   /* into the u32 pool */
    uint32_t nupd, upd_off;         /* into the update pool */
    uint32_t own_seg;               /* with own_src: the image's only segment ... */
    const uint8_t *own_src;         /* ... is read from here, not from the arena (or NULL) */
};

struct GlzCopyDesc {
    const uint8_t *src;
    uint8_t *dst;
    uint64_t bytes;
};

enum { GLZ_DEV_ERR_NO_SPACE = -3, GLZ_DEV_ERR_BAD_TYPE = -4 };

#endif
