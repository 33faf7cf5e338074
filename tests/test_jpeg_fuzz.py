"""Seeded random JPEG frames through the batch interface: sizes from 1x1 up, all
four bitmap formats, qualities 1..100, both DCTs, both row orders, content from
flat to noise (noise at quality 100 makes the longest codes and ZRL runs).
Against the CPU restatement of libjpeg, which the golden vectors pin to real
libjpeg."""
import numpy as np
import pytest

from spice_amd import binding as B
from spice_amd import frames

from helpers import assert_same_stream
from test_jpeg_parity import batch_encode

FMT_O = {B.BITMAP_FMT_16BIT: 6, B.BITMAP_FMT_24BIT: 7, B.BITMAP_FMT_32BIT: 8, B.BITMAP_FMT_RGBA: 9}


def random_frame(rng, w, h):
    kind = rng.integers(0, 5)
    im = np.zeros((h, w, 4), np.uint8)
    if kind == 0:
        im[..., :3] = rng.integers(0, 256, (h, w, 3))
    elif kind == 1:
        im[..., :3] = rng.integers(0, 256, 3)
    elif kind == 2:
        im[..., 0] = (np.arange(w) * 255 // max(1, w - 1))[None, :]
        im[..., 1] = (np.arange(h) * 255 // max(1, h - 1))[:, None]
        im[..., 2] = rng.integers(0, 256)
    elif kind == 3:
        im[..., :3] = rng.integers(0, 2, (h, w, 1)) * 255          # black / white salt
    else:
        im = frames.golden_image(max(w, 16), max(h, 16), "desktop", int(rng.integers(0, 99)))[:h, :w].copy()
    return im


def run_fuzz(dev, oracle, seed, n, max_w, max_h):
    rng = np.random.default_rng(seed)
    items, wants = [], []
    for _ in range(n):
        w, h = int(rng.integers(1, max_w + 1)), int(rng.integers(1, max_h + 1))
        fmt = [B.BITMAP_FMT_32BIT, B.BITMAP_FMT_RGBA, B.BITMAP_FMT_24BIT, B.BITMAP_FMT_16BIT][int(rng.integers(0, 4))]
        im = random_frame(rng, w, h)
        if fmt == B.BITMAP_FMT_24BIT:
            px = frames.to_bgr24(im)
        elif fmt == B.BITMAP_FMT_16BIT:
            px = frames.to_rgb16(im, rng)
        else:
            px = im.reshape(h, w * 4)
        q = int(rng.choice([1, 20, 50, 70, 85, 95, 100]))
        dct = int(rng.integers(0, 2))
        sign = 1 if rng.random() < 0.8 else -1
        items.append((px, w, h, fmt, q, dct, sign))
        rows = px if sign > 0 else np.ascontiguousarray(px[::-1])
        wants.append(oracle.jpeg_encode(rows, w, h, px.shape[1], FMT_O[fmt], q, dct))
    for k, (it, want, got) in enumerate(zip(items, wants, batch_encode(dev, items))):
        assert_same_stream(want, got, "seed %d frame %d: %dx%d fmt %d q %d dct %d sign %d"
                           % (seed, k, it[1], it[2], it[3], it[4], it[5], it[6]))


@pytest.mark.parametrize("seed", range(3))
def test_sim_jpeg_fuzz(sim, oracle, seed):
    run_fuzz(sim, oracle, seed, 12, 70, 40)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(50, 58))
def test_gpu_jpeg_fuzz(gpu, oracle, seed):
    run_fuzz(gpu, oracle, seed, 60, 900, 500)
