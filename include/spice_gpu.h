/*
 * libspicegpu -- C ABI of the MI355X image-compression path of the SPICE
 * display server.
 *
 * Section 1 re-exports, with the reference's own names, argument meaning and
 * error behaviour, the encoder-library entry points that red_worker.c calls,
 * so that libspice-server can link this library instead of compiling
 * glz_encoder.c / glz_encoder_dictionary.c / mjpeg_encoder.c / jpeg_encoder.c.
 * Host pointers in, host buffers out, synchronous.
 *
 * Section 2 is the device-resident batch interface the throughput figures are
 * measured on: pixels already in HBM, compressed bytes left in HBM, many
 * sessions / frames per call.
 *
 * No type from HIP or any other runtime appears here: device memory is a
 * plain pointer, a stream is a void *.
 */
#ifndef SPICE_GPU_H
#define SPICE_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ======================================================================
 * Section 0 -- library state
 * ====================================================================== */

/* Selects the GPU this process encodes on and creates the library's stream.
 * Returns 0, or -1 (see spice_gpu_last_error) e.g. when no GPU is visible.
 * Every other entry point fails the same way when no device is usable: there
 * is no CPU fallback. */
int spice_gpu_init(int device);
void spice_gpu_shutdown(void);
const char *spice_gpu_last_error(void);
/* The library's own HIP stream (a hipStream_t), for callers that want to
 * bracket launches with their own events. */
void *spice_gpu_stream(void);

/* ======================================================================
 * Section 1a -- GLZ, drop-in for server/glz_encoder.h:28-52,
 * server/glz_encoder_dictionary.h:30-67, server/glz_encoder_config.h:24-44
 * ====================================================================== */

/* LzImageType: order fixed by server/red_common.h:39-51 and
 * client/glz_decoder.cpp:212-236 (the enum itself lives in spice-common). */
typedef enum {
    LZ_IMAGE_TYPE_INVALID,
    LZ_IMAGE_TYPE_PLT1_LE,
    LZ_IMAGE_TYPE_PLT1_BE,
    LZ_IMAGE_TYPE_PLT4_LE,
    LZ_IMAGE_TYPE_PLT4_BE,
    LZ_IMAGE_TYPE_PLT8,
    LZ_IMAGE_TYPE_RGB16,
    LZ_IMAGE_TYPE_RGB24,
    LZ_IMAGE_TYPE_RGB32,
    LZ_IMAGE_TYPE_RGBA,
    LZ_IMAGE_TYPE_XXXA,
    LZ_IMAGE_TYPE_A8
} LzImageType;

typedef void GlzEncoderContext;
typedef void GlzEncDictContext;
typedef void GlzEncDictImageContext;
typedef void GlzUsrImageContext;
typedef struct GlzEncoderUsrContext GlzEncoderUsrContext;

/* server/glz_encoder_config.h:27-44 -- same member order, same meaning */
struct GlzEncoderUsrContext {
    void (*error)(GlzEncoderUsrContext *usr, const char *fmt, ...);
    void (*warn)(GlzEncoderUsrContext *usr, const char *fmt, ...);
    void (*info)(GlzEncoderUsrContext *usr, const char *fmt, ...);
    void *(*malloc)(GlzEncoderUsrContext *usr, int size);
    void (*free)(GlzEncoderUsrContext *usr, void *ptr);
    /* next chunk of image rows; returns the number of rows */
    int (*more_lines)(GlzEncoderUsrContext *usr, uint8_t **lines);
    /* next chunk of output space; returns its size in bytes */
    int (*more_space)(GlzEncoderUsrContext *usr, uint8_t **io_ptr);
    /* an image left the window because of the size limit */
    void (*free_image)(GlzEncoderUsrContext *usr, GlzUsrImageContext *image);
};

/* server/glz_encoder_dictionary.h:34-38 */
typedef struct GlzEncDictRestoreData {
    uint32_t size;
    uint32_t max_encoders;
    uint64_t last_image_id;
} GlzEncDictRestoreData;

/* server/glz_encoder_dictionary.h:43-44.  size: window capacity in pixels
 * (at most 2^25).  The window's pixel copies and the 2^20-entry hash table
 * live in HBM. */
GlzEncDictContext *glz_enc_dictionary_create(uint32_t size, uint32_t max_encoders,
                                             GlzEncoderUsrContext *usr);
/* server/glz_encoder_dictionary.h:46 */
void glz_enc_dictionary_destroy(GlzEncDictContext *opaque_dict, GlzEncoderUsrContext *usr);
/* server/glz_encoder_dictionary.h:49 */
uint32_t glz_enc_dictionary_get_size(GlzEncDictContext *opaque_dict);
/* server/glz_encoder_dictionary.h:53-55 */
void glz_enc_dictionary_get_restore_data(GlzEncDictContext *opaque_dict,
                                         GlzEncDictRestoreData *out_data,
                                         GlzEncoderUsrContext *usr);
/* server/glz_encoder_dictionary.h:58-59 */
GlzEncDictContext *glz_enc_dictionary_restore(GlzEncDictRestoreData *restore_data,
                                              GlzEncoderUsrContext *usr);
/* server/glz_encoder_dictionary.h:62 */
void glz_enc_dictionary_reset(GlzEncDictContext *opaque_dict, GlzEncoderUsrContext *usr);
/* server/glz_encoder_dictionary.h:66-67 -- logical removal, no callback */
void glz_enc_dictionary_remove_image(GlzEncDictContext *opaque_dict,
                                     GlzEncDictImageContext *image, GlzEncoderUsrContext *usr);

/* What a dictionary holds in device memory.  No counterpart in the reference,
 * whose window points into guest RAM (server/glz_encoder_dictionary.c:469). */
typedef struct SpiceGpuGlzDictInfo {
    uint64_t arena_bytes;   /* pixel copies of the window's images */
    uint64_t table_bytes;   /* hash table + segment table */
    uint32_t images;        /* images in the window */
    uint32_t segments;      /* segment slots, used and free */
    uint32_t window_pixels; /* pixels of those images */
    uint32_t arena_regrows; /* times the arena was re-made larger */
} SpiceGpuGlzDictInfo;
void spice_gpu_glz_dictionary_info(GlzEncDictContext *opaque_dict, SpiceGpuGlzDictInfo *out);

/* The window's bookkeeping as it stands (SharedDictionary's lists,
 * server/glz_encoder_dictionary_protected.h:80-126), for checkers: writes at
 * most cap words and returns how many the dump has.  Words: size limit, max
 * encoders, last image id, pixels so far, segment slots, used segments head and
 * tail, every encoder's head (0xffffffff = none); the number of images in the
 * window, then per image oldest first: id, size in pixels, alive, usr_context,
 * LzImageType, number of segments, and per segment: slot, pixels_so_far,
 * pixels_num, next slot; at last the free segment slots, a count and then the
 * slots in the order in which they will be handed out. */
size_t spice_gpu_glz_dictionary_dump(GlzEncDictContext *opaque_dict, uint64_t *out, size_t cap);

/* server/glz_encoder.h:30-33 */
GlzEncoderContext *glz_encoder_create(uint8_t id, GlzEncDictContext *dictionary,
                                      GlzEncoderUsrContext *usr);
void glz_encoder_destroy(GlzEncoderContext *opaque_encoder);

/* server/glz_encoder.h:49-52.  Returns the compressed size in bytes.  Rows
 * are pulled through usr->more_lines when lines == NULL / num_lines == 0, one
 * window segment per chunk; output goes to io_ptr and then to the buffers
 * usr->more_space hands out.  Unlike the reference the window keeps its own
 * copy of the pixels (in HBM), so the caller's rows may be released as soon
 * as the call returns; free_image still fires on the reference's schedule. */
int glz_encode(GlzEncoderContext *opaque_encoder, LzImageType type, int width, int height,
               int top_down, uint8_t *lines, unsigned int num_lines, int stride,
               uint8_t *io_ptr, unsigned int num_io_bytes, GlzUsrImageContext *usr_context,
               GlzEncDictImageContext **o_enc_dict_context);

/* ======================================================================
 * Section 1b -- MJPEG, drop-in for server/mjpeg_encoder.h:24-112
 * ====================================================================== */

enum {
    MJPEG_ENCODER_FRAME_UNSUPPORTED = -1,
    MJPEG_ENCODER_FRAME_DROP,
    MJPEG_ENCODER_FRAME_ENCODE_START
};

/* SPICE_BITMAP_FMT_* values used by this path (server/red_worker.c:511-512) */
enum {
    SPICE_GPU_BITMAP_FMT_16BIT = 6,
    SPICE_GPU_BITMAP_FMT_24BIT = 7,
    SPICE_GPU_BITMAP_FMT_32BIT = 8,
    SPICE_GPU_BITMAP_FMT_RGBA = 9
};

typedef struct MJpegEncoder MJpegEncoder;

/* server/mjpeg_encoder.h:40-44 */
typedef struct MJpegEncoderRateControlCbs {
    uint32_t (*get_roundtrip_ms)(void *opaque);
    uint32_t (*get_source_fps)(void *opaque);
    void (*update_client_playback_delay)(void *opaque, uint32_t delay_ms);
} MJpegEncoderRateControlCbs;

/* server/mjpeg_encoder.h:46-50 */
typedef struct MJpegEncoderStats {
    uint64_t starting_bit_rate;
    uint64_t cur_bit_rate;
    double avg_quality;
} MJpegEncoderStats;

/* server/mjpeg_encoder.h:52-54.  With bit_rate_control != 0, cbs must carry
 * get_roundtrip_ms and get_source_fps (server/mjpeg_encoder.c:192); the stream
 * then starts probing from quality 50 at 5 fps.  Without it the quality is a
 * fixed 70 (server/mjpeg_encoder.c:33,213). */
MJpegEncoder *mjpeg_encoder_new(int bit_rate_control, uint64_t starting_bit_rate,
                                MJpegEncoderRateControlCbs *cbs, void *opaque);
void mjpeg_encoder_destroy(MJpegEncoder *encoder);
/* server/mjpeg_encoder.h:56 */
uint8_t mjpeg_encoder_get_bytes_per_pixel(MJpegEncoder *encoder);
/* server/mjpeg_encoder.h:69-72.  format is a SpiceBitmapFmt.  *dest must be
 * NULL or malloc()ed; it may be replaced by a larger malloc()ed buffer. */
int mjpeg_encoder_start_frame(MJpegEncoder *encoder, int format, int width, int height,
                              uint8_t **dest, size_t *dest_len, uint32_t frame_mm_time);
/* server/mjpeg_encoder.h:73-74.  Returns rows accepted (1), 0 on failure. */
int mjpeg_encoder_encode_scanline(MJpegEncoder *encoder, uint8_t *src_pixels, size_t image_width);
/* server/mjpeg_encoder.h:75.  Runs the frame on the GPU; returns the JFIF size. */
size_t mjpeg_encoder_end_frame(MJpegEncoder *encoder);
/* server/mjpeg_encoder.h:94-100: a periodic report from the client about the
 * frames it received (end_frame_delay = end_frame_mm_time - client mm_time;
 * audio_delay = UINT32_MAX when no audio plays). */
void mjpeg_encoder_client_stream_report(MJpegEncoder *encoder, uint32_t num_frames,
                                        uint32_t num_drops, uint32_t start_frame_mm_time,
                                        uint32_t end_frame_mm_time, int32_t end_frame_delay,
                                        uint32_t audio_delay);
/* server/mjpeg_encoder.h:109: a frame was dropped because the pipe was congested */
void mjpeg_encoder_notify_server_frame_drop(MJpegEncoder *encoder);
/* server/mjpeg_encoder.h:111-112 */
uint64_t mjpeg_encoder_get_bit_rate(MJpegEncoder *encoder);
void mjpeg_encoder_get_stats(MJpegEncoder *encoder, MJpegEncoderStats *stats);

/* Inspection of the rate controller, named after the fields of
 * MJpegEncoderRateControl and its members (server/mjpeg_encoder.c:82-153).
 * No counterpart in the reference, whose state is private to the .c file. */
typedef struct SpiceGpuMjpegRateState {
    int32_t during_quality_eval, eval_type, eval_reason;
    int32_t min_quality_id, min_quality_fps, max_quality_id, max_quality_fps;
    int32_t max_sampled_fps, max_sampled_fps_quality_id;
    int32_t quality_id;
    uint32_t fps;
    uint32_t num_recent_enc_frames;
    uint64_t encoded_size_by_quality[7];
    double adjusted_fps;
    uint64_t adjusted_fps_start_time, adjusted_fps_num_frames;
    uint64_t byte_rate, base_enc_size, last_enc_size, sum_recent_enc_size;
    uint64_t warmup_start_time;
    uint64_t change_start_time, last_frame_time, sum_enc_size;
    uint32_t change_start_mm_time, num_enc_frames;
    int32_t was_upgraded, max_video_latency;
    uint32_t max_audio_latency;
    uint32_t num_frames_encoded, num_frames_dropped;
    uint32_t playback_delay_updates, playback_delay_last; /* update_client_playback_delay calls */
    uint32_t reserved;
} SpiceGpuMjpegRateState;
void spice_gpu_mjpeg_rate_state(MJpegEncoder *encoder, SpiceGpuMjpegRateState *out);
/* Replace CLOCK_MONOTONIC (server/mjpeg_encoder.c:204,742,934) by now_ns(opaque);
 * NULL restores it.  A stream still in its 3 s warm-up restarts the warm-up at
 * the new clock's present.  For replaying recorded sessions and for tests. */
void spice_gpu_mjpeg_set_clock(MJpegEncoder *encoder, uint64_t (*now_ns)(void *opaque), void *opaque);
/* A rate controller with no codec behind it: mjpeg_encoder_start_frame on it
 * always fails; frames are fed by size with spice_gpu_mjpeg_rate_feed, which
 * does what start_frame (returns its DROP / ENCODE_START) and, when the frame
 * is admitted, end_frame do to the controller.  size_by_quality_id[k] is the
 * size the frame would have at quality 20 + 10k; the entry of the quality the
 * controller picks is used, and 0 there stands for a frame whose scanlines
 * failed (server/mjpeg_encoder.c:848-851).  Needs no GPU.  *quality receives
 * the JPEG quality the frame is coded with. */
MJpegEncoder *spice_gpu_mjpeg_rate_only_new(uint64_t starting_bit_rate,
                                            MJpegEncoderRateControlCbs *cbs, void *opaque,
                                            uint64_t (*now_ns)(void *opaque), void *clock_opaque);
int spice_gpu_mjpeg_rate_feed(MJpegEncoder *encoder, uint32_t frame_mm_time,
                              const uint64_t *size_by_quality_id, int *quality);

/* ======================================================================
 * Section 1c -- still JPEG, drop-in for server/jpeg_encoder.h:35-60
 * ====================================================================== */

typedef enum {
    JPEG_IMAGE_TYPE_INVALID,
    JPEG_IMAGE_TYPE_RGB16,
    JPEG_IMAGE_TYPE_RGB24,
    JPEG_IMAGE_TYPE_BGR24,
    JPEG_IMAGE_TYPE_BGRX32
} JpegEncoderImageType;

typedef void *JpegEncoderContext;
typedef struct JpegEncoderUsrContext JpegEncoderUsrContext;

/* server/jpeg_encoder.h:48-51 */
struct JpegEncoderUsrContext {
    int (*more_space)(JpegEncoderUsrContext *usr, uint8_t **io_ptr);
    int (*more_lines)(JpegEncoderUsrContext *usr, uint8_t **lines);
};

/* server/jpeg_encoder.h:53-54 */
JpegEncoderContext *jpeg_encoder_create(JpegEncoderUsrContext *usr);
void jpeg_encoder_destroy(JpegEncoderContext *encoder);
/* server/jpeg_encoder.h:58-60.  ISLOW DCT, 4:2:0, standard tables; rows top
 * to bottom, further rows through usr->more_lines, output through io_ptr and
 * then usr->more_space.  Returns the JFIF size, 0 on failure. */
int jpeg_encode(JpegEncoderContext *jpeg, int quality, JpegEncoderImageType type, int width,
                int height, uint8_t *lines, unsigned int num_lines, int stride,
                uint8_t *io_ptr, unsigned int num_io_bytes);

/* ======================================================================
 * Section 2 -- device-resident batch interface (no counterpart in the
 * reference, which encodes one image per call on the worker thread;
 * these are what server/red_worker.c:6003-6102 and :8341-8377 become when
 * many display sessions share one GPU)
 * ====================================================================== */

/* One image for one GLZ encoder.  The image's rows are nchunks runs of whole
 * rows in device memory; each run becomes one window segment, exactly like a
 * SpiceChunk on the host path (server/red_worker.c:5700-5728). */
typedef struct SpiceGpuGlzJob {
    GlzEncoderContext *encoder;
    LzImageType type;
    int width, height, top_down, stride;
    int nchunks;
    const void *const *chunk_ptrs;   /* host array of device pointers */
    const uint32_t *chunk_rows;      /* host array */
    void *out;                       /* device buffer for the compressed stream */
    uint32_t out_capacity;
    GlzUsrImageContext *usr_context;
    /* results */
    int32_t out_size;                /* bytes written, or < 0 on error */
    GlzEncDictImageContext *dict_image;
} SpiceGpuGlzJob;

/* Bytes the output buffer of a job must hold in the worst case. */
size_t spice_gpu_glz_out_bound(LzImageType type, int width, int height);

/* Encodes njobs images in one call: a parse launch with a workgroup per job
 * (256 lanes, eight to a CU; 576 lanes and a CU each when the launch has no more
 * jobs than the device has CUs, see spice_gpu_glz_lone_max_jobs), and a copy of the
 * rows into each dictionary's window in device memory.  Jobs that share a dictionary (the display channels of one client,
 * server/red_worker.c:9677-9754) are encoded one after the other in the
 * caller's job order -- the batch is cut into as many launches as the most
 * images any one dictionary gets -- and produce exactly the streams that
 * separate glz_encode calls in that order would.  Submit and finish must be
 * called by the same thread (the batch holds its dictionaries' locks between the
 * two; a finish from another thread is refused and the batch stays in flight).
 * An image of one chunk is parsed where
 * the caller put it and its copy runs beside the parse; other images are copied
 * first.  The chunk buffers must stay valid and unchanged until the call (or
 * spice_gpu_glz_finish) returns.  Workgroups start in the order of expected
 * cost (image size x how the dictionary's previous image compressed), so a
 * batch may hold more jobs than the device runs at once; results come back in
 * the caller's job order.  Returns 0, or -1 with spice_gpu_last_error set (job
 * fields say which job failed). */
int spice_gpu_glz_encode_batch(SpiceGpuGlzJob *jobs, int njobs);

/* As above but split in two: submit enqueues everything and returns, finish waits
 * and completes the host bookkeeping (post-encode window maintenance).  A thread may
 * have up to FOUR batches in flight, over disjoint dictionaries (a batch that names a
 * dictionary of one of the caller's batches in flight is refused): each runs on streams
 * of its own, so while one batch's slowest images are still being parsed another
 * batch's images take the places its quick ones have left.  finish completes the
 * caller's OLDEST batch and takes that batch's job array; a fifth submit from the same
 * thread is refused, another thread's submit waits for a free place. */
int spice_gpu_glz_submit(SpiceGpuGlzJob *jobs, int njobs);
int spice_gpu_glz_finish(SpiceGpuGlzJob *jobs, int njobs);
/* Host time the last batch's submit + finish took, the wait for the device
 * excluded: window bookkeeping, job staging, result hand-out. */
uint64_t spice_gpu_glz_last_batch_host_ns(void);

/* Two kernels parse GLZ images, with the same streams: one fits eight images on a CU
 * (a full GPU: throughput is the number of sessions it holds), one gives an image a CU
 * of its own and runs the scan of the next window, the walk over the current one's stops
 * and the emit of the previous one side by side (an under-filled GPU: what counts is how
 * soon one image is done).  A launch -- a batch, or one level of a batch whose jobs share
 * dictionaries, or the callers a synchronous glz_encode round collected -- of at most
 * njobs images uses the second.  Default: the device's number of CUs; 0: never; a
 * negative number restores the default.  No counterpart in the reference. */
void spice_gpu_glz_lone_max_jobs(int njobs);

enum { SPICE_GPU_DCT_ISLOW = 0, SPICE_GPU_DCT_IFAST = 1 };

/* One frame / still image for the JPEG kernels. */
typedef struct SpiceGpuJpegJob {
    const void *pixels;   /* device pointer to the first (top) row */
    int format;           /* SPICE_GPU_BITMAP_FMT_* */
    int width, height;
    long stride;          /* bytes between rows; negative walks bottom-up */
    int quality;          /* 1..100, libjpeg scaling */
    int dct;              /* SPICE_GPU_DCT_* */
    void *out;            /* device buffer for the JFIF file */
    uint32_t out_capacity;
    int32_t out_size;     /* result: bytes written, or < 0 */
} SpiceGpuJpegJob;

size_t spice_gpu_jpeg_out_bound(int width, int height);
int spice_gpu_jpeg_encode_batch(SpiceGpuJpegJob *jobs, int njobs);
/* The same in two halves.  A thread may have TWO batches in flight: submit, submit,
 * finish, submit, finish ...  Each runs on a stream of its own with its own scratch;
 * the second batch's strip kernel starts when the first one's has ended, so what
 * overlaps is one batch's short bit-stream kernels, copies and host work with the
 * other's strip kernel (a video server codes frame k+1 of its streams while the bytes
 * of frame k are still being put together).  finish completes the OLDEST batch and
 * takes that batch's job array; the two batches must not share output buffers.  A
 * third submit, a finish whose job count is not the oldest batch's, and
 * spice_gpu_jpeg_encode_batch while batches are in flight are refused; another
 * thread's submit waits until the last batch in flight has been finished. */
int spice_gpu_jpeg_submit(SpiceGpuJpegJob *jobs, int njobs);
int spice_gpu_jpeg_finish(SpiceGpuJpegJob *jobs, int njobs);

/* One frame, like a spice_gpu_jpeg_encode_batch of one job, but without any
 * allocation: no counterpart in the reference; it is for callers that cannot
 * allocate device memory where they call from (tight serving loops, harnesses
 * that forbid it; it waits for its result, so it is not for code under HIP graph
 * capture).  The kernels' scratch (the
 * strips' Huffman bits, their counts and offsets, the unstuffed scan, the job
 * record) lives in an 18 MB array inside the library's code object, one copy per
 * GPU, held until the process ends.  Limits: at most 1024 strips of 10 MCUs
 * (rows of 16 pixels x ceil(ceil(width / 16) / 10); 1920x1088 has 816) and
 * out_capacity <= 4 MiB; beyond them the call fails with a message and the
 * batch interface must be used.  spice_gpu_init must have run.  Calls on one GPU
 * share that GPU's array: the library serialises them, and nothing else may use
 * the result of one call's scratch.  Same output bytes as the batch interface.
 * Returns 0 with job->out_size set, or -1. */
int spice_gpu_jpeg_encode_noalloc(SpiceGpuJpegJob *job);

/* ---- Section 3: what decides between the codecs ------------------------
 *
 * Graduality of a bitmap, _get_bitmap_graduality_level
 * (server/red_worker.c:5904-5967) over red_bitmap_utils.h:43-147: every 15th
 * pixel (14th when the width divides 15) of all rows but a chunk's last is
 * compared with its right, lower and lower-right neighbours; a pair scores
 * 0.5 (equal), 1.0 (a channel differs by >= 60, or >= 8 of 31 for 16 bpp) or
 * -0.25; squares of four equal pixels count 0.  The mean over 3 x samples is
 * compared with -0.03 (0 for 16 bpp) and 0.002.  Rows are taken as `width`
 * pixels apart, as the reference does (it scores only bitmaps whose stride has
 * no padding). */
typedef enum {
    BITMAP_GRADUAL_INVALID,
    BITMAP_GRADUAL_NOT_AVAIL,
    BITMAP_GRADUAL_LOW,
    BITMAP_GRADUAL_MEDIUM,
    BITMAP_GRADUAL_HIGH,
} BitmapGradualType; /* server/red_worker.c:798-804 */

#define SPICE_GPU_GRADUAL_MAX_CHUNKS 16

typedef struct SpiceGpuGradualJob {
    int format;                 /* SPICE_GPU_BITMAP_FMT_16BIT / 24BIT / 32BIT / RGBA */
    int width;
    int num_chunks;             /* <= SPICE_GPU_GRADUAL_MAX_CHUNKS */
    const void *chunks[SPICE_GPU_GRADUAL_MAX_CHUNKS];   /* device pointers */
    uint32_t chunk_rows[SPICE_GPU_GRADUAL_MAX_CHUNKS];
    /* results */
    int level;                  /* BitmapGradualType */
    double score;               /* the reference's score, bit for bit */
    int64_t score_quarters;     /* sum of the pair scores in units of 0.25 */
    int64_t num_samples;
} SpiceGpuGradualJob;

/* Bitmaps resident in HBM, any number per call (one kernel launch). */
int spice_gpu_graduality_batch(SpiceGpuGradualJob *jobs, int njobs);

/* The reference's call shape: chunks in host memory.  Returns the level, or
 * BITMAP_GRADUAL_INVALID on error. */
BitmapGradualType spice_gpu_bitmap_graduality_level(int format, int width, int num_chunks,
                                                    const uint8_t *const *chunks,
                                                    const uint32_t *chunk_rows);

/* red_compress_image's choice of codec (server/red_worker.c:6411-6525) as a
 * pure function of what it looks at. */
typedef enum {                  /* spice_image_compression_t, server/spice.h:469-477 */
    SPICE_IMAGE_COMPRESS_INVALID = 0,
    SPICE_IMAGE_COMPRESS_OFF = 1,
    SPICE_IMAGE_COMPRESS_AUTO_GLZ = 2,
    SPICE_IMAGE_COMPRESS_AUTO_LZ = 3,
    SPICE_IMAGE_COMPRESS_QUIC = 4,
    SPICE_IMAGE_COMPRESS_GLZ = 5,
    SPICE_IMAGE_COMPRESS_LZ = 6,
} spice_image_compression_t;

typedef enum {
    SPICE_GPU_CODEC_NONE = 0,           /* send the bitmap as it is (the reference returns FALSE) */
    SPICE_GPU_CODEC_QUIC = 1,
    SPICE_GPU_CODEC_JPEG = 2,
    SPICE_GPU_CODEC_GLZ = 3,
    SPICE_GPU_CODEC_LZ = 4,
    SPICE_GPU_CODEC_NEEDS_GRADUALITY = -1, /* call again with the bitmap's graduality filled in */
    SPICE_GPU_CODEC_ERROR = -2,
} SpiceGpuCodec;

typedef struct SpiceGpuCodecQuery {
    int image_compression;      /* spice_image_compression_t of the worker */
    int format;                 /* SPICE_GPU_BITMAP_FMT_* (SPICE_BITMAP_FMT_* numbering) */
    uint32_t x, y, stride;
    int chunks_unstable;        /* SPICE_CHUNKS_FLAGS_UNSTABLE */
    int can_lossy;
    int enable_jpeg;            /* display_channel->enable_jpeg */
    int graduality;             /* drawable->copy_bitmap_graduality; BITMAP_GRADUAL_INVALID = not known */
    uint32_t glz_dict_size;     /* glz_enc_dictionary_get_size() */
    int glz_frozen;             /* dcc->glz_dict->migrate_freeze */
} SpiceGpuCodecQuery;

SpiceGpuCodec spice_gpu_choose_codec(const SpiceGpuCodecQuery *q);

/* Plain device-memory helpers so that hosts without a HIP binding (the C
 * server, ctypes tests) can stage inputs and fetch results. */
void *spice_gpu_malloc(size_t bytes);
void spice_gpu_free(void *dev_ptr);
int spice_gpu_upload(void *dev_dst, const void *host_src, size_t bytes);
int spice_gpu_download(void *host_dst, const void *dev_src, size_t bytes);

/* Per-codec counters in the shape of stat_info_t / print_compress_stats
 * (server/red_worker.c:183-193, 1142-1215). */
typedef struct SpiceGpuCodecStats {
    uint64_t count;
    uint64_t orig_bytes;
    uint64_t comp_bytes;
    uint64_t total_ns;
    uint64_t min_ns;
    uint64_t max_ns;
} SpiceGpuCodecStats;
enum { SPICE_GPU_STAT_GLZ = 0, SPICE_GPU_STAT_MJPEG = 1, SPICE_GPU_STAT_JPEG = 2, SPICE_GPU_STAT_COUNT = 3 };
void spice_gpu_get_stats(int codec, SpiceGpuCodecStats *out);
void spice_gpu_reset_stats(void);

/* Kernel timing (no counterpart in the reference): when enabled, every kernel
 * the library launches is bracketed by HIP events on the stream it is launched
 * on, and the accumulated device time per kernel name can be read back once the
 * batches have been finished.  total_ms adds the launches' durations up;
 * union_ms is the length of the union of their intervals, which is what the
 * device spent on the kernel when launches of two batches in flight overlap
 * (equal to total_ms when they do not). */
typedef struct SpiceGpuKernelTime {
    char name[48];
    uint64_t launches;
    double total_ms;
    double union_ms;
} SpiceGpuKernelTime;
void spice_gpu_timing_enable(int on);
void spice_gpu_timing_reset(void);
int spice_gpu_timing_read(SpiceGpuKernelTime *out, int max_entries);

#ifdef __cplusplus
}
#endif
#endif
