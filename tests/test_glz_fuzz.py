"""Seeded random GLZ sessions: image sizes, pixel types, content classes, chunk
splits, row order, interface (glz_encode or the device-resident batch) and
logical removals all drawn at random, windows small enough that images are
evicted and the arena is re-made.  Every stream must equal the oracle's."""
import numpy as np
import pytest

from spice_amd import binding as B

from helpers import GlzPair, pixels_for, small_desktop, split_rows

TYPES = [B.LZ_IMAGE_TYPE_RGB32, B.LZ_IMAGE_TYPE_RGBA, B.LZ_IMAGE_TYPE_RGB24, B.LZ_IMAGE_TYPE_RGB16]


def content(rng, w, h, history):
    kind = rng.integers(0, 8)
    im = np.zeros((h, w, 4), np.uint8)
    if kind == 0:                                   # noise
        im[..., :3] = rng.integers(0, 256, (h, w, 3))
    elif kind == 1:                                 # one colour
        im[..., :3] = rng.integers(0, 256, 3)
    elif kind == 2:                                 # horizontal gradient with steps
        step = int(rng.integers(1, 9))
        im[..., :3] = ((np.arange(w) // step) % 256)[None, :, None]
    elif kind == 3:                                 # short runs of few colours
        pal = rng.integers(0, 256, (int(rng.integers(2, 6)), 3))
        runs = rng.geometric(1.0 / float(rng.choice([1.5, 4, 20, 300])), size=w * h)
        idx = np.repeat(rng.integers(0, len(pal), size=runs.size), runs)[: w * h]
        im.reshape(-1, 4)[:, :3] = pal[idx]
    elif kind == 4:                                 # periodic
        period = int(rng.choice([2, 3, 7, 31, 63, 64, 65, 130, 257]))
        base = rng.integers(0, 256, (period, 3))
        im.reshape(-1, 4)[:, :3] = base[np.arange(w * h) % period]
    elif kind == 5 and history:                     # an earlier picture again, shifted by a few pixels
        old = history[int(rng.integers(0, len(history)))]
        flat = np.resize(old.reshape(-1, 4), (w * h, 4))
        im = np.roll(flat, int(rng.integers(-5, 6)), axis=0).reshape(h, w, 4).copy()
    elif kind == 6 and history:                     # an earlier picture with a few pixels changed
        old = history[-1]
        im = np.resize(old.reshape(-1, 4), (w * h, 4)).reshape(h, w, 4).copy()
        for _ in range(int(rng.integers(1, 12))):
            im[int(rng.integers(0, h)), int(rng.integers(0, w)), :3] = rng.integers(0, 256, 3)
    else:                                           # desktop-like
        im = small_desktop(w, h, int(rng.integers(0, 50)))
    return im


def run_session(dev, oracle, seed, nimages, max_w, max_h, windows=(3000, 20000, 120000)):
    rng = np.random.default_rng(seed)
    typ = TYPES[seed % 4]
    window = int(rng.choice(windows))
    pair = GlzPair(dev, oracle, window, batch=bool(seed % 2))
    history = []
    try:
        for i in range(nimages):
            for _ in range(20):
                w, h = int(rng.integers(1, max_w + 1)), int(rng.integers(1, max_h + 1))
                if w * h <= window:
                    break
            else:
                w, h = 8, 8
            im = content(rng, w, h, history)
            history = (history + [im])[-4:]
            px = pixels_for(typ, im, rng)
            nchunks = int(rng.integers(1, 5))
            _, oimg, gimg = pair.encode(typ, w, h, px.shape[1], split_rows(px, nchunks),
                                        top_down=bool(rng.integers(0, 2)),
                                        what="fuzz seed %d image %d (%dx%d)" % (seed, i, w, h))
            if rng.random() < 0.08:                 # the newest image is certainly still in the window
                pair.remove(oimg, gimg, 1000 + pair.n - 1)
    finally:
        pair.close()


@pytest.mark.parametrize("seed", range(8))
def test_sim_fuzz(sim, oracle, seed):
    run_session(sim, oracle, seed, 30, 160, 48)


@pytest.mark.gpu
@pytest.mark.usefixtures("both_glz_kernels")     # both GLZ parse kernels (tests/conftest.py)
@pytest.mark.parametrize("seed", range(100, 124))
def test_gpu_fuzz(gpu, oracle, seed):
    run_session(gpu, oracle, seed, 120, 700, 300)
    run_session(gpu, oracle, seed + 1000, 60, 1600, 900, windows=(400000, 1500000, 4000000))
