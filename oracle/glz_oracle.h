/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference GLZ encoder,
 * its shared dictionary/window, and the client-side GLZ decoder.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker.  The product (libspicegpu.so)
 * never links or calls it.
 *
 * PARITY STATUS: "parity unpinned" for encoder *choices*.  The reference GLZ
 * encoder cannot be built in this image (server/glz_encoder.h:24 and
 * server/glz_encoder_config.h:21-22 include common/lz_common.h and
 * spice/macros.h from the un-vendored spice-common / spice-protocol trees) and
 * the reference holds no golden bitstreams for it.  What IS pinned:
 *   - the 33-byte header layout, against the header bytes of a reference run
 *     recorded when the reference was first studied (tests/golden/glz_header.json);
 *   - the wire format, by decoding every oracle stream with a restatement of
 *     the in-tree client decoder (client/glz_decode_tmpl.c:179-319,
 *     client/glz_decoder.cpp:51-105,238-301) and comparing pixels.
 *
 * Constants that live in the absent common/lz_common.h (enum order, MAX_COPY,
 * magic, version, per-type tables) are restated from their in-tree uses:
 * server/red_common.h:39-51, client/glz_decoder.cpp:70-71,212-236,
 * client/glz_decode_tmpl.c:193-196.
 */
#ifndef GLZ_ORACLE_H
#define GLZ_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    GLZO_TYPE_INVALID = 0,
    GLZO_TYPE_PLT1_LE,
    GLZO_TYPE_PLT1_BE,
    GLZO_TYPE_PLT4_LE,
    GLZO_TYPE_PLT4_BE,
    GLZO_TYPE_PLT8,
    GLZO_TYPE_RGB16,
    GLZO_TYPE_RGB24,
    GLZO_TYPE_RGB32,
    GLZO_TYPE_RGBA,
    GLZO_TYPE_XXXA,
    GLZO_TYPE_A8
};

enum {
    GLZO_ERR_STRIDE = -1,      /* glz_encoder.c:236-247 */
    GLZO_ERR_TOO_BIG = -2,     /* glz_encoder_dictionary.c:397-399 */
    GLZO_ERR_NO_SPACE = -3,    /* glz_encoder.c:64-67 */
    GLZO_ERR_BAD_TYPE = -4,    /* glz_encoder.c:299-302 */
    GLZO_ERR_LINES = -5,       /* glz_encoder_dictionary.c:497-501 */
    GLZO_ERR_DECODE = -6
};

typedef struct GlzoDict GlzoDict;
typedef struct GlzoDecWindow GlzoDecWindow;

typedef struct GlzoStats {
    uint64_t stops;        /* iterations of the compress_seg main loop */
    uint64_t literals;     /* literal pixels emitted */
    uint64_t matches;      /* dictionary matches emitted */
    uint64_t rle_matches;  /* run matches emitted */
    uint64_t match_pixels; /* pixels covered by matches of both kinds */
} GlzoStats;

GlzoDict *glzo_dict_create(uint32_t size_px, uint32_t max_encoders);
GlzoDict *glzo_dict_restore(uint32_t size_px, uint32_t max_encoders, uint64_t last_image_id);
void glzo_dict_get_restore_data(GlzoDict *d, uint32_t *size_px, uint32_t *max_encoders,
                                uint64_t *last_image_id);
void glzo_dict_reset(GlzoDict *d);
void glzo_dict_destroy(GlzoDict *d);
uint32_t glzo_dict_get_size(GlzoDict *d);

/* Encodes one image given as nchunks chunks of whole rows (each chunk becomes
 * one window segment, glz_encoder_dictionary.c:504-525).  Pixel data is copied,
 * so the caller's buffers may be released after the call.  Returns the number
 * of bytes written to out, or a negative GLZO_ERR_*. */
long glzo_encode(GlzoDict *d, uint32_t encoder_id, int type, int width, int height,
                 int top_down, int stride, const uint8_t *const *chunks,
                 const uint32_t *chunk_rows, int nchunks, uint8_t *out, size_t out_cap,
                 uint64_t usr_tag, void **o_image);

void glzo_remove_image(GlzoDict *d, void *image);
/* usr tags of images whose free_image callback fired since the last call */
int glzo_pop_freed(GlzoDict *d, uint64_t *tags, int max);
void glzo_get_stats(GlzoDict *d, GlzoStats *out, int reset);

/* Record every parser stop as a pair (position in its segment, length:
 * 0 literal, bit 31 run, bit 30 match into another image).  Process-wide. */
void glzo_set_trace(uint32_t *buf, size_t cap_pairs);
size_t glzo_trace_count(void);

GlzoDecWindow *glzo_decwin_create(void);
void glzo_decwin_destroy(GlzoDecWindow *w);
/* Decodes one GLZ stream to its own pixel type.  Returns bytes consumed from
 * `in`, or GLZO_ERR_DECODE.  out must hold gross_pixels * bytes-per-pixel. */
long glzo_decode(GlzoDecWindow *w, const uint8_t *in, size_t in_len, uint8_t *out,
                 size_t out_cap, int *o_type, int *o_width, int *o_height, int *o_top_down,
                 uint64_t *o_id, uint32_t *o_win_head_dist);

#ifdef __cplusplus
}
#endif
#endif
