"""TEST INFRASTRUCTURE ONLY -- ctypes bindings for the CPU oracle.

Loaded by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, as
the checker; the product package (spice_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

TYPE_PLT8, TYPE_RGB16, TYPE_RGB24, TYPE_RGB32, TYPE_RGBA = 5, 6, 7, 8, 9
BPP = {TYPE_PLT8: 1, TYPE_RGB16: 2, TYPE_RGB24: 3, TYPE_RGB32: 4, TYPE_RGBA: 4}
DCT_ISLOW, DCT_IFAST = 0, 1
COLOR_V6B, COLOR_IJG9 = 0, 1
FMT_16BIT, FMT_24BIT, FMT_32BIT, FMT_RGBA = 6, 7, 8, 9

_lib = None
_ref = None


def build(force=False):
    """Compiles liboracle.so (and _ref/libjpegref.so where libjpeg headers exist)."""
    so = os.path.join(HERE, "liboracle.so")
    srcs = ["glz_oracle.c", "glz_oracle.h", "glz_oracle_parse.inc", "jpeg_oracle.c",
            "jpeg_oracle.h", "gradual_oracle.c", "mjpeg_rate_oracle.c", "ref_jpeg_driver.c",
            "Makefile"]
    newest = max(os.path.getmtime(os.path.join(HERE, s)) for s in srcs)
    if force or not os.path.exists(so) or os.path.getmtime(so) < newest:
        subprocess.check_call(["make", "-s", "-C", HERE, "all"])
    return so


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        L.glzo_dict_create.restype = C.c_void_p
        L.glzo_dict_create.argtypes = [C.c_uint32, C.c_uint32]
        L.glzo_dict_restore.restype = C.c_void_p
        L.glzo_dict_restore.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64]
        L.glzo_dict_get_restore_data.argtypes = [C.c_void_p, C.POINTER(C.c_uint32),
                                                 C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]
        L.glzo_dict_reset.argtypes = [C.c_void_p]
        L.glzo_dict_destroy.argtypes = [C.c_void_p]
        L.glzo_dict_get_size.restype = C.c_uint32
        L.glzo_dict_get_size.argtypes = [C.c_void_p]
        L.glzo_encode.restype = C.c_long
        L.glzo_encode.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.c_int,
                                  C.c_void_p, C.c_size_t, C.c_uint64, C.POINTER(C.c_void_p)]
        L.glzo_remove_image.argtypes = [C.c_void_p, C.c_void_p]
        L.glzo_pop_freed.restype = C.c_int
        L.glzo_pop_freed.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
        L.glzo_get_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64 * 5), C.c_int]
        L.glzo_decwin_create.restype = C.c_void_p
        L.glzo_decwin_destroy.argtypes = [C.c_void_p]
        L.glzo_decode.restype = C.c_long
        L.glzo_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                  C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                  C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]
        L.jpgo_encode.restype = C.c_long
        L.jpgo_encode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_int, C.c_int,
                                  C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        L.jpgo_header.restype = C.c_long
        L.jpgo_header.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        L.jpgo_coefficients.restype = C.c_long
        L.jpgo_coefficients.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_int, C.c_int,
                                        C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        L.grado_level.restype = C.c_int
        L.grado_level.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32),
                                  C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        L.grado_choose.restype = C.c_int
        L.grado_choose.argtypes = [C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_uint32, C.c_int]
        L.mjro_new.restype = C.c_void_p
        L.mjro_new.argtypes = [C.c_uint64, C.c_uint64, C.c_int]
        L.mjro_free.argtypes = [C.c_void_p]
        L.mjro_env.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint32]
        L.mjro_feed.restype = C.c_int
        L.mjro_feed.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
        L.mjro_client_report.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                         C.c_int32, C.c_uint32]
        L.mjro_server_drop.argtypes = [C.c_void_p]
        L.mjro_state.argtypes = [C.c_void_p, C.c_void_p]
        L.mjro_asserted.restype = C.c_int
        L.mjro_asserted.argtypes = [C.c_void_p]
        L.mjro_bit_rate.restype = C.c_uint64
        L.mjro_bit_rate.argtypes = [C.c_void_p]
        L.mjro_avg_quality.restype = C.c_double
        L.mjro_avg_quality.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def ref():
    """The real-libjpeg driver, or None when it was not built in this tree."""
    global _ref
    if _ref is None:
        so = os.path.join(HERE, "_ref", "libjpegref.so")
        if not os.path.exists(so):
            return None
        try:
            R = C.CDLL(so)
        except OSError:
            return None
        R.jref_encode.restype = C.c_long
        R.jref_encode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_int, C.c_int,
                                  C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        R.jref_mjpeg_new.restype = C.c_void_p
        R.jref_mjpeg_free.argtypes = [C.c_void_p]
        R.jref_mjpeg_frame.restype = C.c_long
        R.jref_mjpeg_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_long, C.c_int,
                                       C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        _ref = R
    return _ref


class GlzDict:
    """One shared GLZ dictionary of the oracle."""

    def __init__(self, size_px, max_encoders=1, restore_last_id=None):
        L = lib()
        if restore_last_id is None:
            self.h = L.glzo_dict_create(size_px, max_encoders)
        else:
            self.h = L.glzo_dict_restore(size_px, max_encoders, restore_last_id)
        if not self.h:
            raise ValueError("dictionary size above the 2^25 pixel cap")

    def close(self):
        if self.h:
            lib().glzo_dict_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def reset(self):
        lib().glzo_dict_reset(self.h)

    def restore_data(self):
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint64()
        lib().glzo_dict_get_restore_data(self.h, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def encode(self, img_type, width, height, stride, chunks, top_down=True, encoder_id=0,
               usr_tag=0, out_cap=None):
        """chunks: list of contiguous uint8 arrays, each a whole number of rows."""
        L = lib()
        chunks = [np.ascontiguousarray(c, dtype=np.uint8).reshape(-1) for c in chunks]
        n = len(chunks)
        ptrs = (C.c_void_p * n)(*[c.ctypes.data for c in chunks])
        rows = (C.c_uint32 * n)(*[c.size // stride for c in chunks])
        total = sum(c.size for c in chunks)
        if out_cap is None:
            out_cap = total * 2 + 4096
        out = np.empty(out_cap, dtype=np.uint8)
        img = C.c_void_p()
        r = L.glzo_encode(self.h, encoder_id, img_type, width, height, 1 if top_down else 0,
                          stride, ptrs, rows, n, out.ctypes.data, out_cap, usr_tag, C.byref(img))
        if r < 0:
            raise RuntimeError("glz oracle error %d" % r)
        return out[:r].copy(), img.value

    def remove_image(self, handle):
        lib().glzo_remove_image(self.h, handle)

    def pop_freed(self):
        buf = (C.c_uint64 * 4096)()
        n = lib().glzo_pop_freed(self.h, buf, 4096)
        return [buf[i] for i in range(n)]

    def stats(self, reset=False):
        a = (C.c_uint64 * 5)()
        lib().glzo_get_stats(self.h, C.byref(a), 1 if reset else 0)
        return dict(stops=a[0], literals=a[1], matches=a[2], rle_matches=a[3], match_pixels=a[4])


class GlzDecoder:
    """Window of decoded images plus the restated client decoder."""

    def __init__(self):
        self.h = lib().glzo_decwin_create()

    def close(self):
        if self.h:
            lib().glzo_decwin_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def decode(self, stream, out_cap):
        stream = np.ascontiguousarray(stream, dtype=np.uint8)
        out = np.zeros(out_cap, dtype=np.uint8)
        t, w, h, td = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        iid, hd = C.c_uint64(), C.c_uint32()
        r = lib().glzo_decode(self.h, stream.ctypes.data, stream.size, out.ctypes.data, out_cap,
                              C.byref(t), C.byref(w), C.byref(h), C.byref(td), C.byref(iid),
                              C.byref(hd))
        if r < 0:
            raise RuntimeError("glz decode error %d" % r)
        return dict(used=r, type=t.value, width=w.value, height=h.value, top_down=td.value,
                    id=iid.value, win_head_dist=hd.value, pixels=out)


def jpeg_encode(pixels, width, height, stride, fmt, quality, dct, color=COLOR_V6B):
    pixels = np.ascontiguousarray(pixels, dtype=np.uint8)
    cap = width * height * 4 + 65536
    out = np.empty(cap, dtype=np.uint8)
    n = lib().jpgo_encode(pixels.ctypes.data, width, height, stride, fmt, quality, dct, color,
                          out.ctypes.data, cap)
    if n < 0:
        raise RuntimeError("jpeg oracle failed")
    return out[:n].copy()


def jpeg_coefficients(pixels, width, height, stride, fmt, quality, dct, color=COLOR_V6B):
    pixels = np.ascontiguousarray(pixels, dtype=np.uint8)
    nm = ((width + 15) // 16) * ((height + 15) // 16)
    out = np.empty((nm, 6, 64), dtype=np.int16)
    n = lib().jpgo_coefficients(pixels.ctypes.data, width, height, stride, fmt, quality, dct,
                                color, out.ctypes.data, nm)
    if n != nm:
        raise RuntimeError("jpeg oracle failed")
    return out


def jpeg_header(width, height, quality):
    out = np.empty(1024, dtype=np.uint8)
    n = lib().jpgo_header(width, height, quality, out.ctypes.data, 1024)
    return out[:n].copy()


def ref_jpeg_encode(pixels, width, height, stride, fmt, quality, dct, fancy=0):
    R = ref()
    if R is None:
        raise RuntimeError("oracle/_ref/libjpegref.so not built")
    pixels = np.ascontiguousarray(pixels, dtype=np.uint8)
    cap = width * height * 4 + 65536
    out = np.empty(cap, dtype=np.uint8)
    n = R.jref_encode(pixels.ctypes.data, width, height, stride, fmt, quality, dct, fancy,
                      out.ctypes.data, cap)
    if n < 0:
        raise RuntimeError("libjpeg reference failed")
    return out[:n].copy()


# ------------------------------------------------------------- codec choice
def graduality(fmt, width, chunks):
    """(level, score, samples) of a bitmap given as a list of (rows, width * bpp) uint8 arrays."""
    chunks = [np.ascontiguousarray(c, dtype=np.uint8) for c in chunks]
    n = len(chunks)
    ptrs = (C.c_void_p * n)(*[c.ctypes.data for c in chunks])
    rows = (C.c_uint32 * n)(*[c.shape[0] for c in chunks])
    score, samples = C.c_double(0), C.c_int64(0)
    lvl = lib().grado_level(fmt, width, n, ptrs, rows, C.byref(score), C.byref(samples))
    return lvl, score.value, samples.value


def choose_codec(comp, fmt, x, y, stride, unstable, can_lossy, enable_jpeg, graduality_level,
                 glz_dict_size, glz_frozen):
    return lib().grado_choose(comp, fmt, x, y, stride, int(unstable), int(can_lossy), int(enable_jpeg),
                              graduality_level, glz_dict_size, int(glz_frozen))


class MjpegRateState(C.Structure):
    """MJpegEncoderRateControl's fields (server/mjpeg_encoder.c:82-153), flattened."""
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

    def as_dict(self):
        d = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            d[name] = list(v) if name == "encoded_size_by_quality" else v
        return d


class MjpegRate:
    """The reference's rate controller with clock and callbacks as inputs
    (server/mjpeg_encoder.c:362-727, 886-1258)."""

    def __init__(self, starting_bit_rate, now_ns, has_delay_cb=True):
        self.L = lib()
        self.h = self.L.mjro_new(starting_bit_rate, now_ns, 1 if has_delay_cb else 0)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.mjro_free(self.h)
            self.h = None

    def env(self, now_ns, roundtrip_ms, source_fps):
        self.L.mjro_env(self.h, now_ns, roundtrip_ms, source_fps)

    def feed(self, mm_time, size_by_quality_id):
        """-> (verdict, quality): verdict 0 drop, 1 coded, -1 the reference would have aborted.
        size_by_quality_id: the frame's size at each of the 7 qualities (0 = scanline failure)."""
        q = C.c_int(0)
        r = self.L.mjro_feed(self.h, mm_time, (C.c_uint64 * 7)(*size_by_quality_id), C.byref(q))
        return r, q.value

    def client_report(self, num_frames, num_drops, start_mm, end_mm, end_delay, audio_delay):
        self.L.mjro_client_report(self.h, num_frames, num_drops, start_mm, end_mm, end_delay, audio_delay)

    def server_drop(self):
        self.L.mjro_server_drop(self.h)

    def state(self):
        st = MjpegRateState()
        self.L.mjro_state(self.h, C.byref(st))
        return st.as_dict()

    def asserted(self):
        return bool(self.L.mjro_asserted(self.h))

    def bit_rate(self):
        return self.L.mjro_bit_rate(self.h)

    def avg_quality(self):
        return self.L.mjro_avg_quality(self.h)
