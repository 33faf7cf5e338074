"""GLZ: the HIP parse against the CPU oracle, byte for byte.

The same cases run twice: through the CPU wave simulator build of the kernel
sources (no GPU needed; guards every device buffer) and, marked gpu, through
libspicegpu.so on the MI355X.  Both go through the C ABI (glz_encode with the
reference's more_lines / more_space callbacks).
"""
import numpy as np
import pytest

from spice_amd import binding as B
from spice_amd import frames

from helpers import GlzPair, pixels_for, small_desktop, split_rows

TYPES = [B.LZ_IMAGE_TYPE_RGB32, B.LZ_IMAGE_TYPE_RGBA, B.LZ_IMAGE_TYPE_RGB24, B.LZ_IMAGE_TYPE_RGB16]
SIZES = [(64, 32), (100, 60), (7, 3), (1, 1), (3, 1), (200, 50), (2, 2), (130, 70), (6, 1), (7, 1)]

# every test here runs through both GLZ parse kernels (tests/conftest.py)
pytestmark = pytest.mark.usefixtures("both_glz_kernels")


def run_mixed_sequence(dev, oracle, typ, nimages, window=40000, batch=False):
    """Small window so that images are evicted; ragged sizes; 1..3 chunks;
    one logical removal; both row orders."""
    rng = np.random.default_rng(typ)
    pair = GlzPair(dev, oracle, window, batch=batch)
    try:
        for i in range(nimages):
            w, h = SIZES[i % len(SIZES)]
            px = pixels_for(typ, small_desktop(w, h, i // 2), rng)
            chunks = split_rows(px, 1 if i % 3 else 3)
            _, oimg, gimg = pair.encode(typ, w, h, px.shape[1], chunks, top_down=bool(i % 2),
                                        what="mixed", direct=(i % 4 == 1))
            if i == 5:
                pair.remove(oimg, gimg, 1000 + pair.n - 1)
    finally:
        pair.close()


def run_noise_and_runs(dev, oracle, typ, batch=False):
    """Incompressible noise (all literals), a constant image (one run), stripes
    with period inside and outside the 64-position window."""
    rng = np.random.default_rng(99)
    pair = GlzPair(dev, oracle, 1 << 16, batch=batch)
    try:
        w, h = 96, 40
        imgs = []
        noise = np.zeros((h, w, 4), np.uint8)
        noise[..., :3] = rng.integers(0, 256, (h, w, 3))
        imgs.append(noise)
        flat = np.zeros((h, w, 4), np.uint8)
        flat[..., :3] = (9, 200, 77)
        imgs.append(flat)
        for period in (2, 3, 5, 31, 64, 65, 97, 200):
            a = np.zeros((h * w, 4), np.uint8)
            base = rng.integers(0, 256, (period, 3))
            a[:, :3] = base[np.arange(h * w) % period]
            imgs.append(a.reshape(h, w, 4))
        imgs.append(noise)  # cross-image match of the whole frame
        for k, im in enumerate(imgs):
            px = pixels_for(typ, im, rng)
            pair.encode(typ, w, h, px.shape[1], [px], what="pattern %d" % k)
    finally:
        pair.close()


def run_desktop_frames(dev, oracle, w, h, nframes, window, nchunks=1):
    seq = frames.DesktopSequence(w, h, seed=11)
    rng = np.random.default_rng(0)
    pair = GlzPair(dev, oracle, window)
    try:
        for k in range(nframes):
            px = pixels_for(B.LZ_IMAGE_TYPE_RGB32, seq.next(), rng)
            pair.encode(B.LZ_IMAGE_TYPE_RGB32, w, h, w * 4, split_rows(px, nchunks), what="desktop")
    finally:
        pair.close()


def run_plumbing(dev, oracle):
    """The reference harness's own update bitmaps on a 1024x768 primary
    (server/tests/test_display_base.c:212-271)."""
    pair = GlzPair(dev, oracle, 1 << 20)
    try:
        for t in range(6):
            tile = frames.plumbing_tile(1024, t, color=t % 2, c_i=t // 2)
            h, w = tile.shape[:2]
            pair.encode(B.LZ_IMAGE_TYPE_RGB32, w, h, w * 4, [tile.reshape(h, -1)], what="plumbing tile")
        solid = frames.plumbing_solid(320, 240, 0x00ff8040)
        pair.encode(B.LZ_IMAGE_TYPE_RGB32, 320, 240, 320 * 4, [solid.reshape(240, -1)], what="solid")
    finally:
        pair.close()


def run_low_entropy(dev, oracle, typ, nimages, w, h, batch=False):
    """Few colours in short random runs: nearly every 3-pixel hash recurs many
    times inside a 256-position window, so table slots are overwritten by
    stops, by match tails and -- for skipped positions -- not at all; runs of
    every length from 3 pixels to several windows."""
    rng = np.random.default_rng(1000 + typ)
    pair = GlzPair(dev, oracle, w * h * 3, batch=batch)
    try:
        for i in range(nimages):
            ncol = 2 + i % 4
            palette = rng.integers(0, 256, (ncol, 3))
            mean_run = (1.5, 3.0, 8.0, 40.0, 700.0)[i % 5]
            runs = rng.geometric(1.0 / mean_run, size=w * h)
            idx = np.repeat(rng.integers(0, ncol, size=runs.size), runs)[: w * h]
            im = np.zeros((h * w, 4), np.uint8)
            im[:, :3] = palette[idx]
            if i % 3 == 2:  # a periodic stretch: matches at distance < 64 that go on for long
                im[w * 3 : w * 9] = im[w * 3 + np.arange(w * 6) % 37]
            px = pixels_for(typ, im.reshape(h, w, 4), rng)
            pair.encode(typ, w, h, px.shape[1], split_rows(px, 1 + i % 2), what="low entropy %d" % i)
    finally:
        pair.close()


# ------------------------------------------------------------------ CPU (sim)
@pytest.mark.parametrize("typ", TYPES)
def test_sim_mixed_sequence(sim, oracle, typ):
    run_mixed_sequence(sim, oracle, typ, 40)


@pytest.mark.parametrize("typ", [B.LZ_IMAGE_TYPE_RGB32, B.LZ_IMAGE_TYPE_RGBA, B.LZ_IMAGE_TYPE_RGB16])
def test_sim_noise_and_runs(sim, oracle, typ):
    run_noise_and_runs(sim, oracle, typ)


@pytest.mark.parametrize("typ", TYPES)
def test_sim_low_entropy(sim, oracle, typ):
    run_low_entropy(sim, oracle, typ, 10, 173, 29)


@pytest.mark.parametrize("typ", TYPES)
def test_sim_batch_interface_same_cases(sim, oracle, typ):
    """the same images through spice_gpu_glz_encode_batch: single-chunk images are
    parsed straight from the uploaded buffer (which the simulator ends at a guard
    page), the others from the window arena"""
    run_mixed_sequence(sim, oracle, typ, 40, batch=True)
    run_low_entropy(sim, oracle, typ, 10, 173, 29, batch=True)
    if typ != B.LZ_IMAGE_TYPE_RGB24:
        run_noise_and_runs(sim, oracle, typ, batch=True)


def test_sim_desktop_frames(sim, oracle):
    run_desktop_frames(sim, oracle, 320, 200, 4, window=200000, nchunks=2)


def test_sim_plumbing(sim, oracle):
    run_plumbing(sim, oracle)


# ------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("typ", TYPES)
def test_gpu_batch_interface_same_cases(gpu, oracle, typ):
    run_mixed_sequence(gpu, oracle, typ, 60, batch=True)
    run_low_entropy(gpu, oracle, typ, 15, 1031, 67, batch=True)


@pytest.mark.gpu
@pytest.mark.parametrize("typ", TYPES)
def test_gpu_mixed_sequence(gpu, oracle, typ):
    run_mixed_sequence(gpu, oracle, typ, 60)


@pytest.mark.gpu
@pytest.mark.parametrize("typ", TYPES)
def test_gpu_noise_and_runs(gpu, oracle, typ):
    run_noise_and_runs(gpu, oracle, typ)


@pytest.mark.gpu
@pytest.mark.parametrize("typ", TYPES)
def test_gpu_low_entropy(gpu, oracle, typ):
    run_low_entropy(gpu, oracle, typ, 20, 640, 97)


@pytest.mark.gpu
def test_gpu_desktop_frames_small(gpu, oracle):
    run_desktop_frames(gpu, oracle, 640, 400, 6, window=1 << 20, nchunks=3)


@pytest.mark.gpu
def test_gpu_plumbing(gpu, oracle):
    run_plumbing(gpu, oracle)


@pytest.mark.gpu
def test_gpu_desktop_1080p_window_16m(gpu, oracle):
    """BASELINE configs[1]: 1920x1080 frames through a 16 Mi-pixel window; the
    9th frame evicts the first."""
    run_desktop_frames(gpu, oracle, 1920, 1080, 10, window=16 * 1024 * 1024)
