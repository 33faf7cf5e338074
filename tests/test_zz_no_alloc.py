"""spice_gpu_jpeg_encode_noalloc, the entry point that allocates nothing: its
scratch is an array inside the library's code object.  Two calls in a row, the
second with a smaller frame (the array keeps the first call's contents), each
compared byte for byte with spice_gpu_jpeg_encode_batch on the same frame; then
sizes beyond its maximum are refused.  That nothing is allocated is checked too: on
the simulator by counting hipMalloc / hipHostMalloc calls, on the GPU by the free
device memory before and after."""
import ctypes

import numpy as np
import pytest

from spice_amd import binding as B
from spice_amd import frames


def allocation_probe(dev):
    """a function returning a number that changes when the library allocates"""
    if hasattr(dev.L, "sim_alloc_calls"):
        dev.L.sim_alloc_calls.restype = ctypes.c_ulong
        return lambda: int(dev.L.sim_alloc_calls())
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemGetInfo.argtypes = [ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]

    def free_bytes():
        free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
        assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
        return int(free.value)
    return free_bytes


def both_ways(dev, w, h, kind, seed, quality, dct):
    img = frames.golden_image(w, h, kind, seed).reshape(h, w * 4)
    px = dev.to_device(img)
    cap = min(4 << 20, dev.L.spice_gpu_jpeg_out_bound(w, h))
    out_a, out_b = dev.malloc(cap), dev.malloc(cap)
    try:
        item = dict(pixels=px, format=B.BITMAP_FMT_32BIT, width=w, height=h, stride=w * 4,
                    quality=quality, dct=dct, out_capacity=cap)
        n_b = B.jpeg_encode_batch(dev, [dict(item, out=out_b)])[0]
        probe = allocation_probe(dev)
        before = probe()
        n_a = B.jpeg_encode_noalloc(dev, dict(item, out=out_a))
        assert probe() == before, "the entry point that allocates nothing allocated"
        a, b = dev.download(out_a, n_a), dev.download(out_b, n_b)
        assert n_a == n_b and (a == b).all(), "%dx%d %s q%d" % (w, h, kind, quality)
        assert a[0] == 0xff and a[1] == 0xd8 and a[-2] == 0xff and a[-1] == 0xd9
    finally:
        for p in (px, out_a, out_b):
            dev.free(p)


def run(dev, big, small):
    both_ways(dev, big[0], big[1], "desktop", 3, 70, B.DCT_IFAST)
    both_ways(dev, small[0], small[1], "noise", 4, 85, B.DCT_ISLOW)      # smaller, after the larger
    both_ways(dev, small[0] + 13, small[1] + 5, "smooth", 5, 70, B.DCT_IFAST)
    # beyond the maximum: 2048 strips of 10 MCUs
    px = dev.malloc(64)
    out = dev.malloc(64)
    try:
        with pytest.raises(B.SpiceGpuError, match="exceed the maximum"):
            B.jpeg_encode_noalloc(dev, dict(pixels=px, format=B.BITMAP_FMT_32BIT, width=160 * 16, height=128 * 16,
                                            stride=160 * 64, quality=70, dct=B.DCT_IFAST, out=out,
                                            out_capacity=64))
        with pytest.raises(B.SpiceGpuError, match="exceed the maximum"):
            B.jpeg_encode_noalloc(dev, dict(pixels=px, format=B.BITMAP_FMT_32BIT, width=16, height=16,
                                            stride=64, quality=70, dct=B.DCT_IFAST, out=out,
                                            out_capacity=(4 << 20) + 1))
    finally:
        dev.free(px)
        dev.free(out)


def test_sim_jpeg_noalloc(sim):
    run(sim, (200, 120), (64, 40))


@pytest.mark.gpu
def test_gpu_jpeg_noalloc(gpu):
    run(gpu, (1920, 1080), (640, 360))
