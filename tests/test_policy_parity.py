"""Codec choice: the graduality score on the device against the CPU oracle
(integer sums, sample counts, the double score bit for bit, the level), and
red_compress_image's decision table against its restatement.

As everywhere, the cases run on the CPU wave simulator build and, marked gpu, on
the MI355X; both go through the C ABI."""
import itertools

import numpy as np
import pytest

from spice_amd import binding as B
from spice_amd import frames

from helpers import pixels_for, split_rows

FORMATS = [B.BITMAP_FMT_16BIT, B.BITMAP_FMT_24BIT, B.BITMAP_FMT_32BIT, B.BITMAP_FMT_RGBA]
LZ_OF = {B.BITMAP_FMT_16BIT: B.LZ_IMAGE_TYPE_RGB16, B.BITMAP_FMT_24BIT: B.LZ_IMAGE_TYPE_RGB24,
         B.BITMAP_FMT_32BIT: B.LZ_IMAGE_TYPE_RGB32, B.BITMAP_FMT_RGBA: B.LZ_IMAGE_TYPE_RGBA}


def pictures(w, h, rng):
    """text-like, photo-like (smooth + noise), flat, hard noise"""
    yy, xx = np.mgrid[0:h, 0:w]
    smooth = np.zeros((h, w, 4), np.uint8)
    smooth[..., 0] = (xx * 255 // max(1, w - 1)) & 0xff
    smooth[..., 1] = (yy * 255 // max(1, h - 1)) & 0xff
    smooth[..., 2] = ((xx + yy) // 2) & 0xff
    smooth[..., :3] += rng.integers(0, 3, (h, w, 3), dtype=np.uint8)
    flat = np.zeros((h, w, 4), np.uint8)
    flat[..., :3] = (10, 200, 30)
    noise = np.zeros((h, w, 4), np.uint8)
    noise[..., :3] = rng.integers(0, 256, (h, w, 3))
    text = frames.DesktopSequence(max(w, 96), max(h, 96), seed=5).next().reshape(max(h, 96), max(w, 96), 4)[:h, :w]
    return {"smooth": smooth, "flat": flat, "noise": noise, "text": np.ascontiguousarray(text)}


def check_bitmaps(dev, oracle, sizes, on_device):
    rng = np.random.default_rng(3)
    levels = {}
    for (w, h), fmt in itertools.product(sizes, FORMATS):
        for name, im in pictures(w, h, rng).items():
            px = pixels_for(LZ_OF[fmt], im, rng)            # rows of width * bpp bytes
            for nchunks in (1, 3):
                chunks = split_rows(px, nchunks)
                want = oracle.graduality(fmt, w, chunks)
                if on_device:
                    ptrs = [(dev.to_device(c), c.shape[0]) for c in chunks]
                    got = B.graduality_batch(dev, [dict(format=fmt, width=w, chunks=ptrs)])[0]
                    for p, _ in ptrs:
                        dev.free(p)
                    assert (got[0], got[2]) == (want[0], want[2]), (w, h, fmt, name, nchunks, got, want)
                    assert np.float64(got[1]).tobytes() == np.float64(want[1]).tobytes()
                else:
                    assert B.bitmap_graduality_level(dev, fmt, w, chunks) == want[0]
                levels[(name, fmt)] = want[0]
    return levels


SMALL = [(96, 40), (15, 9), (30, 7), (2, 2), (1, 5), (7, 1), (3, 50), (14, 20), (16, 3), (131, 33)]


def test_sim_graduality(sim, oracle):
    lv = check_bitmaps(sim, oracle, SMALL, on_device=True)
    assert lv[("smooth", B.BITMAP_FMT_32BIT)] == B.BITMAP_GRADUAL_HIGH
    assert lv[("text", B.BITMAP_FMT_32BIT)] == B.BITMAP_GRADUAL_LOW


def test_sim_graduality_host_chunks(sim, oracle):
    check_bitmaps(sim, oracle, [(96, 40), (15, 9)], on_device=False)


def test_sim_batch_of_many(sim, oracle):
    rng = np.random.default_rng(8)
    items, want, held = [], [], []
    for k in range(9):
        w, h = SMALL[k % len(SMALL)]
        im = list(pictures(w, h, rng).values())[k % 4]
        px = pixels_for(B.LZ_IMAGE_TYPE_RGB32, im, rng)
        p = sim.to_device(px)
        held.append(p)
        items.append(dict(format=B.BITMAP_FMT_32BIT, width=w, chunks=[(p, h)]))
        want.append(oracle.graduality(B.BITMAP_FMT_32BIT, w, [px]))
    got = B.graduality_batch(sim, items)
    for g, o in zip(got, want):
        assert (g[0], g[1], g[2]) == o
    for p in held:
        sim.free(p)


def test_choose_codec_matches_the_reference_table(sim, oracle):
    """Every combination red_compress_image distinguishes."""
    geoms = [(100, 100, 400), (100, 100, 416), (2, 100, 8), (100, 2, 400), (3, 3, 12), (5000, 4000, 20000),
             (13, 1, 52), (14, 1, 56)]
    n = 0
    for comp in range(0, 8):
        for fmt in range(1, 11):
            for (x, y, bs) in geoms:
                bpp = [0, 0, 0, 0, 0, 1, 2, 3, 4, 4, 1][fmt]
                stride = bs if bpp == 0 else (bs // 4) * bpp if bs % 4 == 0 and bs != 416 else x * bpp + 16
                if bpp == 0:
                    stride = bs // 32 + (1 if fmt < 3 else 0)
                for unstable, lossy, jpeg, frozen in itertools.product((0, 1), repeat=4):
                    for grad in range(5):
                        for dict_size in (1 << 20, 16 << 20):
                            a = (comp, fmt, x, y, stride, unstable, lossy, jpeg, grad, dict_size, frozen)
                            assert B.choose_codec(sim, *a) == oracle.choose_codec(*a), a
                            n += 1
    assert n > 50000


@pytest.mark.gpu
def test_gpu_graduality(gpu, oracle):
    check_bitmaps(gpu, oracle, SMALL + [(640, 360), (1920, 1080)], on_device=True)


@pytest.mark.gpu
def test_gpu_graduality_host_chunks(gpu, oracle):
    check_bitmaps(gpu, oracle, [(96, 40), (15, 9), (333, 77)], on_device=False)
