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


def run_alignment(dev, oracle):
    """16- and 24-bit rows that start on 4-byte boundaries take the wide pixel
    loader, the others the per-pixel one; both against the same oracle."""
    rng = np.random.default_rng(77)
    items, wants = [], []
    for w, h in ((72, 40), (71, 41), (70, 33), (64, 32), (37, 18)):
        for fmt in (B.BITMAP_FMT_24BIT, B.BITMAP_FMT_16BIT):
            im = frames.golden_image(max(w, 16), max(h, 16), "desktop", w)[:h, :w].copy()
            px = frames.to_bgr24(im) if fmt == B.BITMAP_FMT_24BIT else frames.to_rgb16(im, rng)
            for sign in (1, -1):
                items.append((px, w, h, fmt, 70, 1, sign))
                rows = px if sign > 0 else np.ascontiguousarray(px[::-1])
                wants.append(oracle.jpeg_encode(rows, w, h, px.shape[1], FMT_O[fmt], 70, 1))
    for it, want, got in zip(items, wants, batch_encode(dev, items)):
        assert_same_stream(want, got, "%dx%d fmt %d sign %d" % (it[1], it[2], it[3], it[6]))


def test_sim_row_alignment(sim, oracle):
    run_alignment(sim, oracle)


@pytest.mark.gpu
def test_gpu_row_alignment(gpu, oracle):
    run_alignment(gpu, oracle)


def run_density_sweep(dev, oracle, qualities):
    """One-strip images (10 MCUs) whose bit counts sweep from a few hundred to
    tens of thousands: the strip-at-once Huffman path, its code-by-code walk for
    blocks over 64 bits, and the block-at-a-time path past the bit ring's size."""
    rng = np.random.default_rng(5)
    items, wants = [], []
    for amp in (2, 6, 16, 40, 100):
        noise = rng.integers(-amp, amp + 1, (32, 176, 3))
        im = np.zeros((32, 176, 4), np.uint8)
        im[..., :3] = np.clip(128 + noise, 0, 255)
        px = im.reshape(32, 176 * 4)
        for q in qualities:
            for dct in (0, 1):
                items.append((px, 176, 32, B.BITMAP_FMT_32BIT, q, dct, 1))
                wants.append(oracle.jpeg_encode(px, 176, 32, px.shape[1], 8, q, dct))
    sizes = set()
    for it, want, got in zip(items, wants, batch_encode(dev, items)):
        assert_same_stream(want, got, "q %d dct %d" % (it[4], it[5]))
        sizes.add(len(want))
    return sizes


def test_sim_density_sweep(sim, oracle):
    sizes = run_density_sweep(sim, oracle, range(4, 100, 12))
    # header ~600 bytes, 132 blocks: from a few bits a block to over 200 (a full
    # strip of 60 such blocks is past the 8192-bit ring)
    assert min(sizes) < 1000 and max(sizes) > 4000, (min(sizes), max(sizes))


@pytest.mark.gpu
def test_gpu_density_sweep(gpu, oracle):
    run_density_sweep(gpu, oracle, range(2, 101, 2))


def extreme_frames(w, h):
    """the inputs that drive the DCT's operands and the quantiser's dividends to their bounds"""
    y, x = np.mgrid[0:h, 0:w]
    out = []
    for name, v in (("black", np.zeros((h, w), np.uint8)), ("white", np.full((h, w), 255, np.uint8)),
                    ("checker1", ((x + y) & 1) * 255), ("columns", (x & 1) * 255), ("rows", (y & 1) * 255),
                    ("checker8", (((x >> 3) + (y >> 3)) & 1) * 255), ("halves", (x < w // 2) * 255),
                    ("checker4", (((x >> 2) + (y >> 2)) & 1) * 255)):
        im = np.zeros((h, w, 4), np.uint8)
        im[..., :3] = np.asarray(v, np.uint8)[..., None]
        out.append((name, im))
    im = np.zeros((h, w, 4), np.uint8)            # saturated primaries against each other: chroma at its bounds
    im[..., 0] = ((x >> 1) & 1) * 255
    im[..., 1] = ((y >> 1) & 1) * 255
    im[..., 2] = (((x + y) >> 2) & 1) * 255
    out.append(("primaries", im))
    return out


def run_extremes(dev, oracle, w=48, h=32):
    """The single-precision DCT and quantiser at the ends of their ranges (full-swing
    patterns at every spatial frequency, quality 1 = divisors at their largest, quality 100 =
    divisor 1 x scale), both DCTs: byte-identical to the integer restatement of libjpeg."""
    items, wants, names = [], [], []
    for name, im in extreme_frames(w, h):
        px = im.reshape(h, w * 4)
        for q in (1, 2, 50, 99, 100):
            for dct in (0, 1):
                items.append((px, w, h, B.BITMAP_FMT_32BIT, q, dct, 1))
                wants.append(oracle.jpeg_encode(px, w, h, w * 4, 8, q, dct))
                names.append("%s q %d dct %d" % (name, q, dct))
    for name, want, got in zip(names, wants, batch_encode(dev, items)):
        assert_same_stream(want, got, name)


def test_sim_extremes(sim, oracle):
    run_extremes(sim, oracle)


@pytest.mark.gpu
def test_gpu_extremes(gpu, oracle):
    run_extremes(gpu, oracle, 208, 112)
