"""ctypes binding of libspicegpu's C ABI (include/spice_gpu.h).

The classes here mirror the reference's encoder-library interface for the
image-compression path -- same entry points, argument meaning and callback
contract as server/glz_encoder.h, server/glz_encoder_dictionary.h,
server/mjpeg_encoder.h and server/jpeg_encoder.h -- so that host code (and the
parity tests) drive the GPU path the way red_worker.c drives the CPU encoders:
rows are pulled through ``more_lines``, output is pushed into 64 KiB buffers
handed out by ``more_space`` (RedCompressBuf, server/red_worker.c:130,504-509).
"""
import ctypes as C

import numpy as np

LZ_IMAGE_TYPE_RGB16, LZ_IMAGE_TYPE_RGB24, LZ_IMAGE_TYPE_RGB32, LZ_IMAGE_TYPE_RGBA = 6, 7, 8, 9
LZ_BPP = {6: 2, 7: 3, 8: 4, 9: 4}
BITMAP_FMT_16BIT, BITMAP_FMT_24BIT, BITMAP_FMT_32BIT, BITMAP_FMT_RGBA = 6, 7, 8, 9
BITMAP_BPP = {6: 2, 7: 3, 8: 4, 9: 4}
DCT_ISLOW, DCT_IFAST = 0, 1
JPEG_IMAGE_TYPE_RGB16, JPEG_IMAGE_TYPE_RGB24, JPEG_IMAGE_TYPE_BGR24, JPEG_IMAGE_TYPE_BGRX32 = 1, 2, 3, 4
MJPEG_ENCODER_FRAME_UNSUPPORTED, MJPEG_ENCODER_FRAME_DROP, MJPEG_ENCODER_FRAME_ENCODE_START = -1, 0, 1
RED_COMPRESS_BUF_SIZE = 64 * 1024  # server/red_worker.c:130

_MSG_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p, C.c_char_p)
_MALLOC_CB = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_int)
_FREE_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)
_LINES_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_void_p))
_SPACE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_void_p))
_FREE_IMAGE_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)


class GlzEncoderUsrContext(C.Structure):
    _fields_ = [("error", _MSG_CB), ("warn", _MSG_CB), ("info", _MSG_CB), ("malloc", _MALLOC_CB),
                ("free", _FREE_CB), ("more_lines", _LINES_CB), ("more_space", _SPACE_CB),
                ("free_image", _FREE_IMAGE_CB)]


class GlzEncDictRestoreData(C.Structure):
    _fields_ = [("size", C.c_uint32), ("max_encoders", C.c_uint32), ("last_image_id", C.c_uint64)]


class JpegEncoderUsrContext(C.Structure):
    _fields_ = [("more_space", _SPACE_CB), ("more_lines", _LINES_CB)]


class SpiceGpuGlzJob(C.Structure):
    _fields_ = [("encoder", C.c_void_p), ("type", C.c_int), ("width", C.c_int), ("height", C.c_int),
                ("top_down", C.c_int), ("stride", C.c_int), ("nchunks", C.c_int),
                ("chunk_ptrs", C.POINTER(C.c_void_p)), ("chunk_rows", C.POINTER(C.c_uint32)),
                ("out", C.c_void_p), ("out_capacity", C.c_uint32), ("usr_context", C.c_void_p),
                ("out_size", C.c_int32), ("dict_image", C.c_void_p)]


GRADUAL_MAX_CHUNKS = 16
BITMAP_GRADUAL_INVALID, BITMAP_GRADUAL_NOT_AVAIL, BITMAP_GRADUAL_LOW, BITMAP_GRADUAL_MEDIUM, \
    BITMAP_GRADUAL_HIGH = range(5)
IMAGE_COMPRESS_OFF, IMAGE_COMPRESS_AUTO_GLZ, IMAGE_COMPRESS_AUTO_LZ, IMAGE_COMPRESS_QUIC, \
    IMAGE_COMPRESS_GLZ, IMAGE_COMPRESS_LZ = range(1, 7)
CODEC_NONE, CODEC_QUIC, CODEC_JPEG, CODEC_GLZ, CODEC_LZ = range(5)
CODEC_NEEDS_GRADUALITY, CODEC_ERROR = -1, -2


class SpiceGpuGradualJob(C.Structure):
    _fields_ = [("format", C.c_int), ("width", C.c_int), ("num_chunks", C.c_int),
                ("chunks", C.c_void_p * GRADUAL_MAX_CHUNKS), ("chunk_rows", C.c_uint32 * GRADUAL_MAX_CHUNKS),
                ("level", C.c_int), ("score", C.c_double), ("score_quarters", C.c_int64),
                ("num_samples", C.c_int64)]


class SpiceGpuCodecQuery(C.Structure):
    _fields_ = [("image_compression", C.c_int), ("format", C.c_int), ("x", C.c_uint32), ("y", C.c_uint32),
                ("stride", C.c_uint32), ("chunks_unstable", C.c_int), ("can_lossy", C.c_int),
                ("enable_jpeg", C.c_int), ("graduality", C.c_int), ("glz_dict_size", C.c_uint32),
                ("glz_frozen", C.c_int)]


class SpiceGpuJpegJob(C.Structure):
    _fields_ = [("pixels", C.c_void_p), ("format", C.c_int), ("width", C.c_int), ("height", C.c_int),
                ("stride", C.c_long), ("quality", C.c_int), ("dct", C.c_int), ("out", C.c_void_p),
                ("out_capacity", C.c_uint32), ("out_size", C.c_int32)]


class SpiceGpuCodecStats(C.Structure):
    _fields_ = [("count", C.c_uint64), ("orig_bytes", C.c_uint64), ("comp_bytes", C.c_uint64),
                ("total_ns", C.c_uint64), ("min_ns", C.c_uint64), ("max_ns", C.c_uint64)]


class SpiceGpuGlzDictInfo(C.Structure):
    _fields_ = [("arena_bytes", C.c_uint64), ("table_bytes", C.c_uint64), ("images", C.c_uint32),
                ("segments", C.c_uint32), ("window_pixels", C.c_uint32), ("arena_regrows", C.c_uint32)]


class SpiceGpuKernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint64), ("total_ms", C.c_double),
                ("union_ms", C.c_double)]


class MJpegEncoderStats(C.Structure):
    _fields_ = [("starting_bit_rate", C.c_uint64), ("cur_bit_rate", C.c_uint64),
                ("avg_quality", C.c_double)]


class MJpegEncoderRateControlCbs(C.Structure):
    """server/mjpeg_encoder.h:40-44"""
    _fields_ = [("get_roundtrip_ms", C.c_void_p), ("get_source_fps", C.c_void_p),
                ("update_client_playback_delay", C.c_void_p)]


class SpiceGpuMjpegRateState(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "during_quality_eval", "eval_type", "eval_reason", "min_quality_id", "min_quality_fps",
        "max_quality_id", "max_quality_fps", "max_sampled_fps", "max_sampled_fps_quality_id",
        "quality_id")] + [
        ("fps", C.c_uint32), ("num_recent_enc_frames", C.c_uint32),
        ("encoded_size_by_quality", C.c_uint64 * 7), ("adjusted_fps", C.c_double),
        ("adjusted_fps_start_time", C.c_uint64), ("adjusted_fps_num_frames", C.c_uint64),
        ("byte_rate", C.c_uint64), ("base_enc_size", C.c_uint64), ("last_enc_size", C.c_uint64),
        ("sum_recent_enc_size", C.c_uint64), ("warmup_start_time", C.c_uint64),
        ("change_start_time", C.c_uint64), ("last_frame_time", C.c_uint64),
        ("sum_enc_size", C.c_uint64), ("change_start_mm_time", C.c_uint32),
        ("num_enc_frames", C.c_uint32), ("was_upgraded", C.c_int32),
        ("max_video_latency", C.c_int32), ("max_audio_latency", C.c_uint32),
        ("num_frames_encoded", C.c_uint32), ("num_frames_dropped", C.c_uint32),
        ("playback_delay_updates", C.c_uint32), ("playback_delay_last", C.c_uint32),
        ("reserved", C.c_uint32)]


_U32_CB = C.CFUNCTYPE(C.c_uint32, C.c_void_p)
_DELAY_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32)
_CLOCK_CB = C.CFUNCTYPE(C.c_uint64, C.c_void_p)


class StreamEnv:
    """What a stream's rate controller asks its owner (the StreamAgent of
    server/red_worker.c:2984-3020): round trip, source frame rate, and where
    playback-delay updates go.  now_ns, when set, replaces CLOCK_MONOTONIC."""

    def __init__(self, roundtrip_ms=0, source_fps=25, now_ns=None, with_delay_cb=True):
        self.roundtrip_ms = roundtrip_ms
        self.source_fps = source_fps
        self.now_ns = now_ns
        self.playback_delays = []
        self._keep = (_U32_CB(lambda _o: self.roundtrip_ms & 0xFFFFFFFF),
                      _U32_CB(lambda _o: self.source_fps & 0xFFFFFFFF),
                      _DELAY_CB(lambda _o, ms: self.playback_delays.append(ms)),
                      _CLOCK_CB(lambda _o: self.now_ns))
        self.cbs = MJpegEncoderRateControlCbs(
            C.cast(self._keep[0], C.c_void_p), C.cast(self._keep[1], C.c_void_p),
            C.cast(self._keep[2], C.c_void_p) if with_delay_cb else None)

    @property
    def clock(self):
        return C.cast(self._keep[3], C.c_void_p) if self.now_ns is not None else None


EXPORTS = [
    "spice_gpu_init", "spice_gpu_shutdown", "spice_gpu_last_error", "spice_gpu_stream",
    "glz_enc_dictionary_create", "glz_enc_dictionary_destroy", "glz_enc_dictionary_get_size",
    "glz_enc_dictionary_get_restore_data", "glz_enc_dictionary_restore", "glz_enc_dictionary_reset",
    "glz_enc_dictionary_remove_image", "spice_gpu_glz_dictionary_info", "spice_gpu_glz_dictionary_dump", "glz_encoder_create", "glz_encoder_destroy", "glz_encode",
    "mjpeg_encoder_new", "mjpeg_encoder_destroy", "mjpeg_encoder_get_bytes_per_pixel",
    "mjpeg_encoder_start_frame", "mjpeg_encoder_encode_scanline", "mjpeg_encoder_end_frame",
    "mjpeg_encoder_get_bit_rate", "mjpeg_encoder_get_stats",
    "mjpeg_encoder_client_stream_report", "mjpeg_encoder_notify_server_frame_drop",
    "spice_gpu_mjpeg_rate_state", "spice_gpu_mjpeg_set_clock", "spice_gpu_mjpeg_rate_only_new",
    "spice_gpu_mjpeg_rate_feed",
    "jpeg_encoder_create", "jpeg_encoder_destroy", "jpeg_encode",
    "spice_gpu_glz_out_bound", "spice_gpu_glz_encode_batch", "spice_gpu_glz_submit",
    "spice_gpu_glz_finish", "spice_gpu_glz_last_batch_host_ns", "spice_gpu_glz_lone_max_jobs", "spice_gpu_jpeg_out_bound", "spice_gpu_jpeg_encode_batch",
    "spice_gpu_jpeg_submit", "spice_gpu_jpeg_finish", "spice_gpu_jpeg_encode_noalloc", "spice_gpu_malloc", "spice_gpu_free",
    "spice_gpu_upload", "spice_gpu_download", "spice_gpu_get_stats", "spice_gpu_reset_stats",
    "spice_gpu_timing_enable", "spice_gpu_timing_reset", "spice_gpu_timing_read",
    "spice_gpu_graduality_batch", "spice_gpu_bitmap_graduality_level", "spice_gpu_choose_codec",
]


class SpiceGpuError(RuntimeError):
    pass


class SpiceGpu:
    """All entry points of one loaded libspicegpu, with prototypes attached."""

    def __init__(self, cdll):
        L = self.L = cdll
        missing = [n for n in EXPORTS if not hasattr(L, n)]
        if missing:
            raise SpiceGpuError("libspicegpu lacks symbols: %s" % ", ".join(missing))
        vp, u32, i = C.c_void_p, C.c_uint32, C.c_int
        L.spice_gpu_init.restype, L.spice_gpu_init.argtypes = i, [i]
        L.spice_gpu_last_error.restype = C.c_char_p
        L.spice_gpu_stream.restype = vp
        L.glz_enc_dictionary_create.restype = vp
        L.glz_enc_dictionary_create.argtypes = [u32, u32, vp]
        L.glz_enc_dictionary_destroy.argtypes = [vp, vp]
        L.glz_enc_dictionary_get_size.restype, L.glz_enc_dictionary_get_size.argtypes = u32, [vp]
        L.glz_enc_dictionary_get_restore_data.argtypes = [vp, C.POINTER(GlzEncDictRestoreData), vp]
        L.glz_enc_dictionary_restore.restype = vp
        L.glz_enc_dictionary_restore.argtypes = [C.POINTER(GlzEncDictRestoreData), vp]
        L.glz_enc_dictionary_reset.argtypes = [vp, vp]
        L.glz_enc_dictionary_remove_image.argtypes = [vp, vp, vp]
        L.spice_gpu_glz_dictionary_info.argtypes = [vp, C.POINTER(SpiceGpuGlzDictInfo)]
        L.spice_gpu_glz_dictionary_dump.restype = C.c_size_t
        L.spice_gpu_glz_dictionary_dump.argtypes = [vp, C.POINTER(C.c_uint64), C.c_size_t]
        L.glz_encoder_create.restype, L.glz_encoder_create.argtypes = vp, [C.c_uint8, vp, vp]
        L.glz_encoder_destroy.argtypes = [vp]
        L.glz_encode.restype = i
        L.glz_encode.argtypes = [vp, i, i, i, i, vp, C.c_uint, i, vp, C.c_uint, vp, C.POINTER(vp)]
        L.mjpeg_encoder_new.restype = vp
        L.mjpeg_encoder_new.argtypes = [i, C.c_uint64, vp, vp]
        L.mjpeg_encoder_destroy.argtypes = [vp]
        L.mjpeg_encoder_get_bytes_per_pixel.restype = C.c_uint8
        L.mjpeg_encoder_get_bytes_per_pixel.argtypes = [vp]
        L.mjpeg_encoder_start_frame.restype = i
        L.mjpeg_encoder_start_frame.argtypes = [vp, i, i, i, C.POINTER(vp), C.POINTER(C.c_size_t), u32]
        L.mjpeg_encoder_encode_scanline.restype = i
        L.mjpeg_encoder_encode_scanline.argtypes = [vp, vp, C.c_size_t]
        L.mjpeg_encoder_end_frame.restype, L.mjpeg_encoder_end_frame.argtypes = C.c_size_t, [vp]
        L.mjpeg_encoder_get_bit_rate.restype, L.mjpeg_encoder_get_bit_rate.argtypes = C.c_uint64, [vp]
        L.mjpeg_encoder_get_stats.argtypes = [vp, C.POINTER(MJpegEncoderStats)]
        L.mjpeg_encoder_client_stream_report.argtypes = [vp, u32, u32, u32, u32, C.c_int32, u32]
        L.mjpeg_encoder_notify_server_frame_drop.argtypes = [vp]
        L.spice_gpu_mjpeg_rate_state.argtypes = [vp, C.POINTER(SpiceGpuMjpegRateState)]
        L.spice_gpu_mjpeg_set_clock.argtypes = [vp, vp, vp]
        L.spice_gpu_mjpeg_rate_only_new.restype = vp
        L.spice_gpu_mjpeg_rate_only_new.argtypes = [C.c_uint64, vp, vp, vp, vp]
        L.spice_gpu_mjpeg_rate_feed.restype = i
        L.spice_gpu_mjpeg_rate_feed.argtypes = [vp, u32, C.POINTER(C.c_uint64), C.POINTER(i)]
        L.jpeg_encoder_create.restype, L.jpeg_encoder_create.argtypes = vp, [vp]
        L.jpeg_encoder_destroy.argtypes = [vp]
        L.jpeg_encode.restype = i
        L.jpeg_encode.argtypes = [vp, i, i, i, i, vp, C.c_uint, i, vp, C.c_uint]
        L.spice_gpu_glz_out_bound.restype = C.c_size_t
        L.spice_gpu_glz_out_bound.argtypes = [i, i, i]
        for n in ("spice_gpu_glz_encode_batch", "spice_gpu_glz_submit", "spice_gpu_glz_finish"):
            getattr(L, n).restype = i
            getattr(L, n).argtypes = [C.POINTER(SpiceGpuGlzJob), i]
        L.spice_gpu_glz_last_batch_host_ns.restype = C.c_uint64
        L.spice_gpu_glz_lone_max_jobs.argtypes = [i]
        L.spice_gpu_jpeg_out_bound.restype = C.c_size_t
        L.spice_gpu_jpeg_out_bound.argtypes = [i, i]
        for n in ("spice_gpu_jpeg_encode_batch", "spice_gpu_jpeg_submit", "spice_gpu_jpeg_finish"):
            getattr(L, n).restype = i
            getattr(L, n).argtypes = [C.POINTER(SpiceGpuJpegJob), i]
        L.spice_gpu_jpeg_encode_noalloc.restype = i
        L.spice_gpu_jpeg_encode_noalloc.argtypes = [C.POINTER(SpiceGpuJpegJob)]
        L.spice_gpu_malloc.restype, L.spice_gpu_malloc.argtypes = vp, [C.c_size_t]
        L.spice_gpu_free.argtypes = [vp]
        L.spice_gpu_upload.restype, L.spice_gpu_upload.argtypes = i, [vp, vp, C.c_size_t]
        L.spice_gpu_download.restype, L.spice_gpu_download.argtypes = i, [vp, vp, C.c_size_t]
        L.spice_gpu_get_stats.argtypes = [i, C.POINTER(SpiceGpuCodecStats)]
        L.spice_gpu_timing_enable.argtypes = [i]
        L.spice_gpu_timing_read.restype = i
        L.spice_gpu_timing_read.argtypes = [C.POINTER(SpiceGpuKernelTime), i]

    # ---- library state
    def init(self, device=0):
        if self.L.spice_gpu_init(device) != 0:
            raise SpiceGpuError(self.last_error())

    def last_error(self):
        return (self.L.spice_gpu_last_error() or b"").decode("utf-8", "replace")

    def stats(self, codec):
        s = SpiceGpuCodecStats()
        self.L.spice_gpu_get_stats(codec, C.byref(s))
        return {k: getattr(s, k) for k, _ in SpiceGpuCodecStats._fields_}

    def kernel_times(self):
        """{kernel name: (launches, total device ms)} since the last timing reset."""
        buf = (SpiceGpuKernelTime * 32)()
        n = self.L.spice_gpu_timing_read(buf, 32)
        return {buf[k].name.decode(): (buf[k].launches, buf[k].total_ms) for k in range(n)}

    def kernel_union_ms(self):
        """{kernel name: device ms covered by at least one launch of it} since the last timing
        reset: launches of two batches in flight overlap, and their durations must not be added."""
        buf = (SpiceGpuKernelTime * 32)()
        n = self.L.spice_gpu_timing_read(buf, 32)
        return {buf[k].name.decode(): buf[k].union_ms for k in range(n)}

    # ---- device memory
    def malloc(self, nbytes):
        p = self.L.spice_gpu_malloc(nbytes)
        if not p:
            raise SpiceGpuError(self.last_error())
        return p

    def free(self, p):
        self.L.spice_gpu_free(p)

    def upload(self, dev, arr):
        arr = np.ascontiguousarray(arr)
        if self.L.spice_gpu_upload(dev, arr.ctypes.data, arr.nbytes) != 0:
            raise SpiceGpuError(self.last_error())

    def download(self, dev, nbytes):
        out = np.empty(nbytes, dtype=np.uint8)
        if nbytes and self.L.spice_gpu_download(out.ctypes.data, dev, nbytes) != 0:
            raise SpiceGpuError(self.last_error())
        return out

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.malloc(max(arr.nbytes, 1))
        self.upload(p, arr)
        return p


class _BufChain:
    """The caller-owned chain of 64 KiB output buffers (RedCompressBuf)."""

    def __init__(self, buf_size=RED_COMPRESS_BUF_SIZE):
        self.buf_size = buf_size
        self.bufs = []

    def first(self):
        self.bufs = [np.empty(self.buf_size, dtype=np.uint8)]
        return self.bufs[0]

    def more(self):
        self.bufs.append(np.empty(self.buf_size, dtype=np.uint8))
        return self.bufs[-1]

    def gather(self, size):
        # marshaller_add_compressed, server/red_worker.c:5221-5233
        out = np.concatenate(self.bufs)[:size] if self.bufs else np.empty(0, np.uint8)
        assert size <= len(self.bufs) * self.buf_size
        return out.copy()


class GlzUsr:
    """A GlzEncoderUsrContext with the callbacks red_worker.c:5578-5794 provides."""

    def __init__(self, buf_size=RED_COMPRESS_BUF_SIZE):
        self.errors, self.warnings, self.freed_images = [], [], []
        self.chunks, self.next_chunk, self.stride = [], 0, 0
        self.out = _BufChain(buf_size)
        self._libc = C.CDLL(None)
        self._libc.malloc.restype, self._libc.malloc.argtypes = C.c_void_p, [C.c_size_t]
        self._libc.free.argtypes = [C.c_void_p]

        def on_error(usr, fmt, msg):
            self.errors.append((msg or fmt or b"").decode("utf-8", "replace"))

        def on_warn(usr, fmt, msg):
            self.warnings.append((msg or fmt or b"").decode("utf-8", "replace"))

        def on_malloc(usr, size):
            return self._libc.malloc(size)

        def on_free(usr, ptr):
            self._libc.free(ptr)

        def on_more_lines(usr, lines):
            # encoder_usr_more_lines, server/red_worker.c:5700-5728
            if self.next_chunk >= len(self.chunks):
                return 0
            c = self.chunks[self.next_chunk]
            if c.size % self.stride:
                return 0
            self.next_chunk += 1
            lines[0] = c.ctypes.data
            return c.size // self.stride

        def on_more_space(usr, io_ptr):
            # encoder_usr_more_space, server/red_worker.c:5656-5668 (bytes for glz)
            b = self.out.more()
            io_ptr[0] = b.ctypes.data
            return b.size

        def on_free_image(usr, image):
            self.freed_images.append(image)

        self._cbs = (_MSG_CB(on_error), _MSG_CB(on_warn), _MSG_CB(on_warn), _MALLOC_CB(on_malloc),
                     _FREE_CB(on_free), _LINES_CB(on_more_lines), _SPACE_CB(on_more_space),
                     _FREE_IMAGE_CB(on_free_image))
        self.ctx = GlzEncoderUsrContext(*self._cbs)

    @property
    def ptr(self):
        return C.addressof(self.ctx)


class GlzDictionary:
    def __init__(self, gpu, size_px, max_encoders=4, usr=None, restore=None):
        self.gpu, self.usr = gpu, usr or GlzUsr()
        if restore is None:
            self.h = gpu.L.glz_enc_dictionary_create(size_px, max_encoders, self.usr.ptr)
        else:
            self.h = gpu.L.glz_enc_dictionary_restore(C.byref(restore), self.usr.ptr)
        if not self.h:
            raise SpiceGpuError("glz_enc_dictionary_create failed: " + gpu.last_error())

    def close(self):
        if self.h:
            self.gpu.L.glz_enc_dictionary_destroy(self.h, self.usr.ptr)
            self.h = None

    def get_size(self):
        return self.gpu.L.glz_enc_dictionary_get_size(self.h)

    def get_restore_data(self):
        d = GlzEncDictRestoreData()
        self.gpu.L.glz_enc_dictionary_get_restore_data(self.h, C.byref(d), self.usr.ptr)
        return d

    def reset(self):
        self.gpu.L.glz_enc_dictionary_reset(self.h, self.usr.ptr)

    def info(self):
        """device memory held by this dictionary (spice_gpu_glz_dictionary_info)"""
        st = SpiceGpuGlzDictInfo()
        self.gpu.L.spice_gpu_glz_dictionary_info(self.h, C.byref(st))
        return {n: getattr(st, n) for n, _ in st._fields_}

    def remove_image(self, image):
        self.gpu.L.glz_enc_dictionary_remove_image(self.h, image, self.usr.ptr)

    def dump(self):
        """the window's bookkeeping (spice_gpu_glz_dictionary_dump) as a dict"""
        n = self.gpu.L.spice_gpu_glz_dictionary_dump(self.h, None, 0)
        buf = (C.c_uint64 * n)()
        assert self.gpu.L.spice_gpu_glz_dictionary_dump(self.h, buf, n) == n
        w = iter(buf)
        d = dict(size_limit=next(w), max_encoders=next(w), last_image_id=next(w), pixels_so_far=next(w),
                 seg_slots=next(w), used_segs_head=next(w), used_segs_tail=next(w))
        d["encoders_heads"] = [next(w) for _ in range(d["max_encoders"])]
        d["images"] = []
        for _ in range(next(w)):
            im = dict(id=next(w), size=next(w), alive=bool(next(w)), usr_context=next(w), type=next(w))
            im["segs"] = [dict(slot=next(w), pixels_so_far=next(w), pixels_num=next(w), next=next(w))
                          for _ in range(next(w))]
            d["images"].append(im)
        d["free_segs"] = [next(w) for _ in range(next(w))]
        return d


class GlzEncoder:
    def __init__(self, gpu, dictionary, enc_id=0, usr=None):
        self.gpu, self.dict, self.usr = gpu, dictionary, usr or dictionary.usr
        self.h = gpu.L.glz_encoder_create(enc_id, dictionary.h, self.usr.ptr)
        if not self.h:
            raise SpiceGpuError("glz_encoder_create failed")

    def close(self):
        if self.h:
            self.gpu.L.glz_encoder_destroy(self.h)
            self.h = None

    def encode(self, img_type, width, height, stride, chunks, top_down=True, usr_context=0,
               first_lines_direct=False):
        """red_glz_compress_image's call (server/red_worker.c:6034-6045): rows come
        through more_lines unless first_lines_direct hands chunk 0 in directly."""
        u = self.usr
        u.chunks = [np.ascontiguousarray(c, dtype=np.uint8).reshape(-1) for c in chunks]
        u.stride = stride
        u.next_chunk = 0
        lines, num_lines = None, 0
        if first_lines_direct:
            lines, num_lines = u.chunks[0].ctypes.data, u.chunks[0].size // stride
            u.next_chunk = 1
        first = u.out.first()
        image = C.c_void_p()
        n_err = len(u.errors)
        size = self.gpu.L.glz_encode(self.h, img_type, width, height, 1 if top_down else 0, lines,
                                     num_lines, stride, first.ctypes.data, first.size, usr_context,
                                     C.byref(image))
        if len(u.errors) > n_err or size <= 0:
            raise SpiceGpuError("glz_encode: " + "; ".join(u.errors[n_err:] or [self.gpu.last_error()]))
        return u.out.gather(size), image.value


class MJpegEncoder:
    """mjpeg_encoder_new/start_frame/encode_scanline/end_frame (server/mjpeg_encoder.h).

    env (a StreamEnv) switches bit-rate control on, as red_display_create_stream
    does when the client supports stream reports (server/red_worker.c:3039-3051).
    rate_only=True makes a controller with no codec behind it (needs no GPU)."""

    def __init__(self, gpu, env=None, starting_bit_rate=0, rate_only=False):
        self.gpu = gpu
        self.env = env
        L = gpu.L
        if rate_only:
            self.h = L.spice_gpu_mjpeg_rate_only_new(starting_bit_rate, C.addressof(env.cbs), None,
                                                     env.clock, None)
        elif env is not None:
            self.h = L.mjpeg_encoder_new(1, starting_bit_rate, C.addressof(env.cbs), None)
            if self.h and env.clock:
                L.spice_gpu_mjpeg_set_clock(self.h, env.clock, None)
        else:
            self.h = L.mjpeg_encoder_new(0, starting_bit_rate, None, None)
        if not self.h:
            raise SpiceGpuError("mjpeg_encoder_new failed: " + gpu.last_error())
        self._dest = C.c_void_p(None)
        self._dest_len = C.c_size_t(0)
        self._libc = C.CDLL(None)
        self._libc.free.argtypes = [C.c_void_p]

    def close(self):
        if self.h:
            self.gpu.L.mjpeg_encoder_destroy(self.h)
            self.h = None
        if self._dest.value:
            self._libc.free(self._dest)
            self._dest = C.c_void_p(None)

    def encode_frame(self, fmt, width, height, rows, stride, mm_time=0):
        """red_marshall_stream_data's start_frame / encode_frame / end_frame sequence
        (server/red_worker.c:8438-8463, 8341-8377): one scanline call per row.
        Returns None when the rate controller drops the frame."""
        L = self.gpu.L
        rows = np.ascontiguousarray(rows, dtype=np.uint8).reshape(-1)
        r = L.mjpeg_encoder_start_frame(self.h, fmt, width, height, C.byref(self._dest),
                                        C.byref(self._dest_len), mm_time)
        if r == MJPEG_ENCODER_FRAME_DROP:
            return None
        if r != MJPEG_ENCODER_FRAME_ENCODE_START:
            raise SpiceGpuError("mjpeg_encoder_start_frame returned %d" % r)
        base = rows.ctypes.data
        for y in range(height):
            if L.mjpeg_encoder_encode_scanline(self.h, base + y * stride, width) == 0:
                raise SpiceGpuError("mjpeg_encoder_encode_scanline failed")
        n = L.mjpeg_encoder_end_frame(self.h)
        if n == 0:
            raise SpiceGpuError("mjpeg_encoder_end_frame: " + self.gpu.last_error())
        return np.ctypeslib.as_array(C.cast(self._dest, C.POINTER(C.c_uint8)), shape=(n,)).copy()

    # ---- bit-rate control (server/mjpeg_encoder.h:94-112)
    def client_stream_report(self, num_frames, num_drops, start_mm, end_mm, end_frame_delay,
                             audio_delay=0xFFFFFFFF):
        self.gpu.L.mjpeg_encoder_client_stream_report(self.h, num_frames, num_drops, start_mm, end_mm,
                                                      end_frame_delay, audio_delay)

    def notify_server_frame_drop(self):
        self.gpu.L.mjpeg_encoder_notify_server_frame_drop(self.h)

    def bit_rate(self):
        return self.gpu.L.mjpeg_encoder_get_bit_rate(self.h)

    def stats(self):
        st = MJpegEncoderStats()
        self.gpu.L.mjpeg_encoder_get_stats(self.h, C.byref(st))
        return {"starting_bit_rate": st.starting_bit_rate, "cur_bit_rate": st.cur_bit_rate,
                "avg_quality": st.avg_quality}

    def rate_state(self):
        st = SpiceGpuMjpegRateState()
        self.gpu.L.spice_gpu_mjpeg_rate_state(self.h, C.byref(st))
        return {n: (list(getattr(st, n)) if n == "encoded_size_by_quality" else getattr(st, n))
                for n, _ in st._fields_}

    def rate_feed(self, mm_time, size_by_quality_id):
        """-> (MJPEG_ENCODER_FRAME_DROP | _ENCODE_START, quality): a frame known only by the
        size it has at each of the 7 qualities"""
        q = C.c_int(0)
        r = self.gpu.L.spice_gpu_mjpeg_rate_feed(self.h, mm_time, (C.c_uint64 * 7)(*size_by_quality_id),
                                                 C.byref(q))
        return r, q.value


class StreamAgent:
    """One video stream of one client: red_marshall_stream_data's frame gate and
    encode_frame's walk over the bitmap's chunks (server/red_worker.c:8315-8502).

    Without rate control the agent itself skips frames that come sooner than
    1/fps after the last one sent (:8423-8431); with it, the encoder decides."""

    MAX_FPS = 30                              # server/red_worker.c: agent->fps = MAX_FPS (:3036)

    def __init__(self, gpu, env=None, starting_bit_rate=0):
        self.enc = MJpegEncoder(gpu, env=env, starting_bit_rate=starting_bit_rate)
        self.rate_control = env is not None
        self.fps = self.MAX_FPS
        self.last_send_time = 0
        self.drops_fps = 0
        self.frames_sent = 0
        self.size_sent = 0

    def close(self):
        self.enc.close()

    @staticmethod
    def _lines(chunks, stride):
        """red_get_image_line (:8315-8339): rows in memory order, chunk after chunk;
        a chunk that does not hold whole rows ends the walk"""
        for c in chunks:
            c = np.ascontiguousarray(c, dtype=np.uint8).reshape(-1)
            for off in range(0, c.size, stride):
                if c.size - off < stride:
                    yield None
                    return
                yield c[off:off + stride]

    def marshall_stream_data(self, fmt, chunks, image_y, stride, src, top_down, mm_time, now_ns):
        """src = (left, top, right, bottom) inside the bitmap.  Returns the JFIF bytes,
        None for a frame skipped or dropped (the call still succeeded, :8429,8449), or
        raises when the reference would return FALSE."""
        left, top, right, bottom = src
        if not self.rate_control and now_ns - self.last_send_time < 10**9 // self.fps:
            self.drops_fps += 1
            return None
        width, height = right - left, bottom - top
        L, e = self.enc.gpu.L, self.enc
        r = L.mjpeg_encoder_start_frame(e.h, fmt, width, height, C.byref(e._dest), C.byref(e._dest_len),
                                        mm_time & 0xFFFFFFFF)
        if r == MJPEG_ENCODER_FRAME_DROP:
            self.drops_fps += 1
            return None
        if r != MJPEG_ENCODER_FRAME_ENCODE_START:
            raise SpiceGpuError("mjpeg_encoder_start_frame returned %d" % r)
        lines = self._lines(chunks, stride)
        skip = top if top_down else image_y - bottom          # encode_frame, :8355-8358
        for _ in range(skip):
            next(lines, None)
        bpp = L.mjpeg_encoder_get_bytes_per_pixel(e.h)
        for _ in range(height):
            row = next(lines, None)
            if row is None:
                raise SpiceGpuError("encode_frame: the bitmap has too few lines")
            if L.mjpeg_encoder_encode_scanline(e.h, row.ctypes.data + left * bpp, width) == 0:
                raise SpiceGpuError("mjpeg_encoder_encode_scanline failed")
        n = L.mjpeg_encoder_end_frame(e.h)
        if n == 0:
            raise SpiceGpuError("mjpeg_encoder_end_frame: " + self.enc.gpu.last_error())
        self.last_send_time = now_ns
        self.frames_sent += 1
        self.size_sent += n
        return np.ctypeslib.as_array(C.cast(e._dest, C.POINTER(C.c_uint8)), shape=(n,)).copy()


class JpegEncoder:
    """jpeg_encoder_create / jpeg_encode (server/jpeg_encoder.h) with red_worker's callbacks."""

    def __init__(self, gpu, buf_size=RED_COMPRESS_BUF_SIZE):
        self.gpu = gpu
        self.out = _BufChain(buf_size)
        self.chunks, self.next_chunk, self.stride, self.reverse = [], 0, 0, False

        def on_more_space(usr, io_ptr):
            b = self.out.more()
            io_ptr[0] = b.ctypes.data
            return b.size

        def on_more_lines(usr, lines):
            # encoder_usr_more_lines incl. the bottom-up walk, server/red_worker.c:5700-5728
            if self.reverse:
                if self.next_chunk < 0:
                    return 0
                c = self.chunks[self.next_chunk]
                self.next_chunk -= 1
                lines[0] = c.ctypes.data + c.size - self.stride
            else:
                if self.next_chunk >= len(self.chunks):
                    return 0
                c = self.chunks[self.next_chunk]
                self.next_chunk += 1
                lines[0] = c.ctypes.data
            return c.size // self.stride

        self._cbs = (_SPACE_CB(on_more_space), _LINES_CB(on_more_lines))
        self.ctx = JpegEncoderUsrContext(*self._cbs)
        self.h = gpu.L.jpeg_encoder_create(C.addressof(self.ctx))
        if not self.h:
            raise SpiceGpuError("jpeg_encoder_create failed")

    def close(self):
        if self.h:
            self.gpu.L.jpeg_encoder_destroy(self.h)
            self.h = None

    def encode(self, quality, img_type, width, height, stride, chunks, top_down=True):
        """red_jpeg_compress_image's call (server/red_worker.c:6243-6262)."""
        self.chunks = [np.ascontiguousarray(c, dtype=np.uint8).reshape(-1) for c in chunks]
        self.stride = stride
        self.reverse = not top_down
        self.next_chunk = len(self.chunks) - 1 if self.reverse else 0
        first = self.out.first()
        n = self.gpu.L.jpeg_encode(self.h, quality, img_type, width, height, None, 0,
                                   -stride if self.reverse else stride, first.ctypes.data, first.size)
        if n <= 0:
            raise SpiceGpuError("jpeg_encode: " + self.gpu.last_error())
        return self.out.gather(n)


def glz_encode_batch(gpu, items, split=False, raise_on_error=True):
    """Device-resident batch: items are dicts with encoder (GlzEncoder), type,
    width, height, stride, top_down, chunks (device pointers), chunk_rows, out
    (device pointer), out_capacity.  Jobs that share a dictionary are encoded in
    the order given.  Returns (sizes, window images); with raise_on_error=False a
    failed batch returns too, the failed jobs with a size <= 0."""
    n = len(items)
    jobs, keep = glz_jobs(items)
    if split:
        rc = gpu.L.spice_gpu_glz_submit(jobs, n)
        if rc == 0:
            rc = gpu.L.spice_gpu_glz_finish(jobs, n)
    else:
        rc = gpu.L.spice_gpu_glz_encode_batch(jobs, n)
    if rc != 0 and raise_on_error:
        raise SpiceGpuError("glz batch: " + gpu.last_error())
    return [jobs[j].out_size for j in range(n)], [jobs[j].dict_image for j in range(n)]


def glz_jobs(items):
    """The SpiceGpuGlzJob array of a batch (and what must stay alive beside it)."""
    n = len(items)
    jobs = (SpiceGpuGlzJob * n)()
    keep = []
    for j, it in enumerate(items):
        nch = len(it["chunks"])
        ptrs = (C.c_void_p * nch)(*it["chunks"])
        rows = (C.c_uint32 * nch)(*it["chunk_rows"])
        keep.append((ptrs, rows))
        jobs[j].encoder = it["encoder"].h
        jobs[j].type = it["type"]
        jobs[j].width, jobs[j].height = it["width"], it["height"]
        jobs[j].top_down, jobs[j].stride = int(it.get("top_down", 1)), it["stride"]
        jobs[j].nchunks = nch
        jobs[j].chunk_ptrs = ptrs
        jobs[j].chunk_rows = rows
        jobs[j].out, jobs[j].out_capacity = it["out"], it["out_capacity"]
        jobs[j].usr_context = it.get("usr_context", 0)
    return jobs, keep


class GlzBatchPlan:
    """A device-resident batch that is submitted again and again: every job keeps
    its encoder, geometry and output buffer, only the chunk pointers change (each
    session's next frame).  Saves rebuilding the job array per call."""

    def __init__(self, gpu, items):
        self.gpu, self.n = gpu, len(items)
        self.jobs = (SpiceGpuGlzJob * self.n)()
        total = sum(len(it["chunks"]) for it in items)
        self.ptrs = np.zeros(total, dtype=np.uint64)
        self.rows = np.zeros(total, dtype=np.uint32)
        off = 0
        for j, it in enumerate(items):
            nch = len(it["chunks"])
            self.ptrs[off:off + nch] = it["chunks"]
            self.rows[off:off + nch] = it["chunk_rows"]
            job = self.jobs[j]
            job.encoder = it["encoder"].h
            job.type = it["type"]
            job.width, job.height = it["width"], it["height"]
            job.top_down, job.stride = int(it.get("top_down", 1)), it["stride"]
            job.nchunks = nch
            job.chunk_ptrs = C.cast(self.ptrs.ctypes.data + 8 * off, C.POINTER(C.c_void_p))
            job.chunk_rows = C.cast(self.rows.ctypes.data + 4 * off, C.POINTER(C.c_uint32))
            job.out, job.out_capacity = it["out"], it["out_capacity"]
            job.usr_context = it.get("usr_context", 0)
            off += nch
        self._sizes = np.ndarray((self.n,), dtype=np.int32, buffer=self.jobs,
                                 offset=SpiceGpuGlzJob.out_size.offset,
                                 strides=(C.sizeof(SpiceGpuGlzJob),))

    def run(self, chunk_ptrs=None):
        """chunk_ptrs: all jobs' chunk device pointers, flattened in job order.
        Returns the compressed sizes (numpy int32, one per job)."""
        if chunk_ptrs is not None:
            self.ptrs[:] = chunk_ptrs
        if self.gpu.L.spice_gpu_glz_encode_batch(self.jobs, self.n) != 0:
            raise SpiceGpuError("glz batch: " + self.gpu.last_error())
        return self._sizes.copy()

    # Two batches in flight: two plans over DISJOINT dictionaries, each submitted while the
    # other still runs (a parse launch lasts as long as its slowest image; the other
    # batch's workgroups take the places the fast images left).  finish() completes the
    # calling thread's oldest batch, so plans are finished in the order they were submitted.
    def submit(self, chunk_ptrs=None):
        if chunk_ptrs is not None:
            self.ptrs[:] = chunk_ptrs
        if self.gpu.L.spice_gpu_glz_submit(self.jobs, self.n) != 0:
            raise SpiceGpuError("glz batch: " + self.gpu.last_error())

    def finish(self):
        if self.gpu.L.spice_gpu_glz_finish(self.jobs, self.n) != 0:
            raise SpiceGpuError("glz batch: " + self.gpu.last_error())
        return self._sizes.copy()


def jpeg_encode_batch(gpu, items):
    """Device-resident batch of frames; items are dicts with pixels (device
    pointer), format, width, height, stride, quality, dct, out, out_capacity."""
    n = len(items)
    jobs = (SpiceGpuJpegJob * n)()
    for j, it in enumerate(items):
        jobs[j].pixels, jobs[j].format = it["pixels"], it["format"]
        jobs[j].width, jobs[j].height, jobs[j].stride = it["width"], it["height"], it["stride"]
        jobs[j].quality, jobs[j].dct = it["quality"], it["dct"]
        jobs[j].out, jobs[j].out_capacity = it["out"], it["out_capacity"]
    if gpu.L.spice_gpu_jpeg_encode_batch(jobs, n) != 0:
        raise SpiceGpuError("jpeg batch: " + gpu.last_error())
    return [jobs[j].out_size for j in range(n)]


def jpeg_encode_noalloc(gpu, item):
    """One frame through spice_gpu_jpeg_encode_noalloc (same item keys as
    jpeg_encode_batch).  Returns the JFIF size."""
    job = SpiceGpuJpegJob()
    job.pixels, job.format = item["pixels"], item["format"]
    job.width, job.height, job.stride = item["width"], item["height"], item["stride"]
    job.quality, job.dct = item["quality"], item["dct"]
    job.out, job.out_capacity = item["out"], item["out_capacity"]
    if gpu.L.spice_gpu_jpeg_encode_noalloc(C.byref(job)) != 0:
        raise SpiceGpuError("jpeg (no allocation): " + gpu.last_error())
    return job.out_size


class JpegBatchPlan:
    """A device-resident JPEG batch that is submitted again and again (a set of
    video streams): geometry, quality and output buffers stay, only the frames'
    device pointers change.  Saves rebuilding the job array per call."""

    def __init__(self, gpu, items, alt_outs=None):
        """alt_outs: a second set of output buffers (device pointers, one per job), needed
        to keep two batches in flight (submit / finish): the batches alternate between the
        items' buffers and these, so one batch's streams are never written while the
        previous batch's are still being produced."""
        self.gpu, self.n = gpu, len(items)
        self.alt_outs = alt_outs
        self.jobs = (SpiceGpuJpegJob * self.n)()
        for j, it in enumerate(items):
            job = self.jobs[j]
            job.pixels, job.format = it["pixels"], it["format"]
            job.width, job.height, job.stride = it["width"], it["height"], it["stride"]
            job.quality, job.dct = it["quality"], it["dct"]
            job.out, job.out_capacity = it["out"], it["out_capacity"]
        view = lambda field, dt: np.ndarray((self.n,), dtype=dt, buffer=self.jobs, offset=field.offset,
                                            strides=(C.sizeof(SpiceGpuJpegJob),))
        self._pixels = view(SpiceGpuJpegJob.pixels, np.uint64)
        self._sizes = view(SpiceGpuJpegJob.out_size, np.int32)

    def run(self, pixels=None):
        """pixels: every job's frame (device pointers, in job order).  Returns the
        JFIF sizes (numpy int32, one per job)."""
        if pixels is not None:
            self._pixels[:] = pixels
        if self.gpu.L.spice_gpu_jpeg_encode_batch(self.jobs, self.n) != 0:
            raise SpiceGpuError("jpeg batch: " + self.gpu.last_error())
        return self._sizes.copy()

    # Two batches in flight (spice_gpu_jpeg_submit twice before the first finish): the
    # plan keeps a second copy of its job array with the second set of output buffers, so
    # that neither the array nor the streams of the batch in flight are touched while the
    # next one runs.  A caller that pipelines reads the streams of batch k (finish returns
    # which set they are in: `last_outs`) before it submits batch k+2.
    def submit(self, pixels=None):
        """Starts the next batch; at most two may be in flight."""
        if not hasattr(self, "_ring"):
            if self.alt_outs is None or len(self.alt_outs) != self.n:
                raise SpiceGpuError("jpeg plan: submit/finish need alt_outs, a second set of output buffers")
            second = (SpiceGpuJpegJob * self.n)()
            C.memmove(second, self.jobs, C.sizeof(self.jobs))
            for j in range(self.n):
                second[j].out = self.alt_outs[j]
            self._ring = [self.jobs, second]
            self._ring_pixels = [self._pixels,
                                 np.ndarray((self.n,), dtype=np.uint64, buffer=second,
                                            offset=SpiceGpuJpegJob.pixels.offset,
                                            strides=(C.sizeof(SpiceGpuJpegJob),))]
            self._ring_sizes = [self._sizes,
                                np.ndarray((self.n,), dtype=np.int32, buffer=second,
                                           offset=SpiceGpuJpegJob.out_size.offset,
                                           strides=(C.sizeof(SpiceGpuJpegJob),))]
            self._head, self._inflight = 0, 0
        if self._inflight == 2:
            raise SpiceGpuError("jpeg plan: two batches are in flight")
        slot = (self._head + self._inflight) & 1
        if pixels is not None:
            self._ring_pixels[slot][:] = pixels
        if self.gpu.L.spice_gpu_jpeg_submit(self._ring[slot], self.n) != 0:
            raise SpiceGpuError("jpeg batch: " + self.gpu.last_error())
        self._inflight += 1

    def finish(self):
        """Completes the oldest batch in flight and returns its JFIF sizes."""
        if not getattr(self, "_inflight", 0):
            raise SpiceGpuError("jpeg plan: nothing in flight")
        slot = self._head
        rc = self.gpu.L.spice_gpu_jpeg_finish(self._ring[slot], self.n)
        self._head ^= 1
        self._inflight -= 1
        if rc != 0:
            raise SpiceGpuError("jpeg batch: " + self.gpu.last_error())
        self.last_outs = [int(self._ring[slot][j].out) for j in range(self.n)] if slot else None
        return self._ring_sizes[slot].copy()

    def in_flight(self):
        return getattr(self, "_inflight", 0)


def graduality_batch(gpu, items):
    """Device-resident bitmaps; items are dicts with format, width, chunks (list of
    (device pointer, rows)).  Returns a list of (level, score, samples, score_quarters)."""
    n = len(items)
    jobs = (SpiceGpuGradualJob * n)()
    for j, it in enumerate(items):
        jobs[j].format, jobs[j].width, jobs[j].num_chunks = it["format"], it["width"], len(it["chunks"])
        for c, (ptr, rows) in enumerate(it["chunks"]):
            jobs[j].chunks[c] = ptr
            jobs[j].chunk_rows[c] = rows
    gpu.L.spice_gpu_graduality_batch.restype = C.c_int
    if gpu.L.spice_gpu_graduality_batch(jobs, n) != 0:
        raise SpiceGpuError("graduality batch: " + gpu.last_error())
    return [(jobs[j].level, jobs[j].score, jobs[j].num_samples, jobs[j].score_quarters) for j in range(n)]


def bitmap_graduality_level(gpu, fmt, width, chunks):
    """The reference's call shape: chunks are (rows, width * bpp) uint8 arrays in host memory."""
    chunks = [np.ascontiguousarray(c, dtype=np.uint8) for c in chunks]
    n = len(chunks)
    ptrs = (C.c_void_p * n)(*[c.ctypes.data for c in chunks])
    rows = (C.c_uint32 * n)(*[c.shape[0] for c in chunks])
    gpu.L.spice_gpu_bitmap_graduality_level.restype = C.c_int
    gpu.L.spice_gpu_bitmap_graduality_level.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lvl = gpu.L.spice_gpu_bitmap_graduality_level(fmt, width, n, ptrs, rows)
    if lvl == BITMAP_GRADUAL_INVALID:
        raise SpiceGpuError("graduality: " + gpu.last_error())
    return lvl


def choose_codec(gpu, comp, fmt, x, y, stride, unstable, can_lossy, enable_jpeg, graduality,
                 glz_dict_size, glz_frozen):
    q = SpiceGpuCodecQuery(comp, fmt, x, y, stride, int(unstable), int(can_lossy), int(enable_jpeg),
                           graduality, glz_dict_size, int(glz_frozen))
    gpu.L.spice_gpu_choose_codec.restype = C.c_int
    return gpu.L.spice_gpu_choose_codec(C.byref(q))


class ImageCompressor:
    """red_compress_image for one display channel client (server/red_worker.c:6411-6525):
    the choice of codec, then red_glz_compress_image (:6003-6102, with the optional
    zlib wrap) or red_jpeg_compress_image (:6182-6279).  QUIC, LZ and the LZ-coded
    alpha plane of JPEG_ALPHA belong to spice-common, which this library does not
    replace: for those the result names the codec, carries no data and is marked
    "delegate" -- the caller runs that encoder itself or sends the bitmap raw; it is
    never to be taken for a compressed image.

    compress() returns None where the reference returns FALSE (the bitmap goes out
    raw), else a dict with "type" (the SPICE_IMAGE_TYPE_* name), "data" and, for
    ZLIB_GLZ_RGB, "glz_data_size"."""

    MIN_GLZ_SIZE_FOR_ZLIB = 100             # server/red_worker.c:133
    FMT_TO_LZ = {BITMAP_FMT_16BIT: LZ_IMAGE_TYPE_RGB16, BITMAP_FMT_24BIT: LZ_IMAGE_TYPE_RGB24,
                 BITMAP_FMT_32BIT: LZ_IMAGE_TYPE_RGB32, BITMAP_FMT_RGBA: LZ_IMAGE_TYPE_RGBA}
    FMT_TO_JPEG = {BITMAP_FMT_16BIT: JPEG_IMAGE_TYPE_RGB16, BITMAP_FMT_24BIT: JPEG_IMAGE_TYPE_BGR24,
                   BITMAP_FMT_32BIT: JPEG_IMAGE_TYPE_BGRX32, BITMAP_FMT_RGBA: JPEG_IMAGE_TYPE_BGRX32}

    def __init__(self, gpu, image_compression, glz_window_px, enable_jpeg=True, jpeg_quality=85,
                 enable_zlib_glz_wrap=False, zlib_level=3):
        self.gpu = gpu
        self.image_compression = image_compression
        self.enable_jpeg, self.jpeg_quality = enable_jpeg, jpeg_quality     # :10623-10624
        self.enable_zlib_glz_wrap, self.zlib_level = enable_zlib_glz_wrap, zlib_level
        self.dict = GlzDictionary(gpu, glz_window_px, 1)
        self.glz = GlzEncoder(gpu, self.dict, 0)
        self.jpeg = JpegEncoder(gpu)
        self.migrate_freeze = False
        self._zlib, self._zlib_level_in_use = None, None

    def close(self):
        self.jpeg.close()
        self.glz.close()
        self.dict.close()

    def _zlib_encode(self, data):
        """zlib_encode (server/zlib_encoder.c:68-125): one deflate stream per image from a
        z_stream that is reset, not re-created; input arrives in 64 KiB RedCompressBufs"""
        import zlib
        # deflateReset keeps the level; Python has no reset, a fresh object is the same stream
        z = zlib.compressobj(self.zlib_level)
        out = [z.compress(bytes(data[o:o + RED_COMPRESS_BUF_SIZE]))
               for o in range(0, len(data), RED_COMPRESS_BUF_SIZE)]
        out.append(z.flush())
        return np.frombuffer(b"".join(out), dtype=np.uint8)

    def compress(self, fmt, x, y, stride, chunks, top_down=True, unstable=False, can_lossy=False,
                 drawable_graduality=BITMAP_GRADUAL_INVALID, usr_context=0):
        query = lambda grad: choose_codec(self.gpu, self.image_compression, fmt, x, y, stride, unstable,
                                          can_lossy, self.enable_jpeg, grad, self.dict.get_size(),
                                          self.migrate_freeze)
        codec = query(drawable_graduality)
        if codec == CODEC_NEEDS_GRADUALITY:          # :6451-6455
            codec = query(bitmap_graduality_level(self.gpu, fmt, x, chunks))
        if codec == CODEC_NONE:
            return None
        if codec == CODEC_ERROR:
            raise SpiceGpuError("invalid image compression type %r" % (self.image_compression,))
        if codec == CODEC_QUIC:
            return {"type": "QUIC", "data": None, "delegate": True}
        if codec == CODEC_LZ:
            return {"type": "LZ_RGB", "data": None, "delegate": True}
        if codec == CODEC_JPEG:
            if fmt not in self.FMT_TO_JPEG:
                return None                          # :6218-6219
            if fmt == BITMAP_FMT_RGBA:
                # the JPEG is followed by lz_encode(XXXA) of the alpha plane in one chain (:6281-6300)
                return {"type": "JPEG_ALPHA", "data": None, "delegate": True}
            data = self.jpeg.encode(self.jpeg_quality, self.FMT_TO_JPEG[fmt], x, y, stride, chunks,
                                    top_down=top_down)
            if len(data) > y * stride:
                return None                          # :6263-6266
            return {"type": "JPEG", "data": data, "is_lossy": True}
        data, image = self.glz.encode(self.FMT_TO_LZ[fmt], x, y, stride, chunks, top_down=top_down,
                                      usr_context=usr_context)
        if self.enable_zlib_glz_wrap and len(data) >= self.MIN_GLZ_SIZE_FOR_ZLIB:
            wrapped = self._zlib_encode(data)
            if len(wrapped) < len(data):             # :6075-6083
                return {"type": "ZLIB_GLZ_RGB", "data": wrapped, "glz_data_size": len(data),
                        "glz_image": image}
        return {"type": "GLZ_RGB", "data": data, "glz_image": image}
