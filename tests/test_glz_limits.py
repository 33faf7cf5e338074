"""Limits and failure behaviour of the GLZ drop-in entry points, against the
reference's rules: window cap (client/red_client.cpp:875,
glz_encoder_dictionary.c:50), "image is bigger than window"
(glz_encoder_dictionary.c:397-399), stride must equal width * bpp
(glz_encoder.c:243-247), more_lines running dry (glz_encoder_dictionary.c:497-501),
more_space refusing (glz_encoder.c:52-73), and that a failed call leaves the
dictionary usable."""
import numpy as np
import pytest

from spice_amd import binding as B

from helpers import GlzPair, pixels_for, small_desktop, split_rows

T32 = B.LZ_IMAGE_TYPE_RGB32


def run_limits(dev, oracle):
    assert not dev.L.glz_enc_dictionary_create((1 << 25) + 1, 1, B.GlzUsr().ptr)
    assert "2^25" in dev.last_error()

    rng = np.random.default_rng(3)
    pair = GlzPair(dev, oracle, 64 * 40)
    try:
        px = pixels_for(T32, small_desktop(64, 40, 1), rng)
        pair.encode(T32, 64, 40, 256, [px], what="exactly the window")   # 2560 px == the limit

        big = pixels_for(T32, small_desktop(64, 41, 2), rng)
        with pytest.raises(RuntimeError):                                 # oracle: GLZO_ERR_TOO_BIG
            pair.od.encode(T32, 64, 41, 256, [big])
        with pytest.raises(B.SpiceGpuError, match="bigger than window"):
            pair.ge.encode(T32, 64, 41, 256, [big])

        with pytest.raises(RuntimeError):                                 # oracle: GLZO_ERR_STRIDE
            pair.od.encode(T32, 60, 40, 256, [px])
        with pytest.raises(B.SpiceGpuError, match="stride"):
            pair.ge.encode(T32, 60, 40, 256, [px])

        with pytest.raises(B.SpiceGpuError, match="palette"):
            pair.ge.encode(5, 64, 40, 64, [px[:, :64]])                   # LZ_IMAGE_TYPE_PLT8

        with pytest.raises(RuntimeError):                                 # rows missing
            pair.od.encode(T32, 64, 40, 256, [px[:30]])
        with pytest.raises(B.SpiceGpuError, match="more lines"):
            pair.ge.encode(T32, 64, 40, 256, [px[:30]])

        # none of the refused calls touched the windows: both sides go on in step
        for k in range(4):
            im = pixels_for(T32, small_desktop(64, 20, k // 2), rng)
            pair.encode(T32, 64, 20, 256, split_rows(im, 1 + k % 2), what="after errors %d" % k)
    finally:
        pair.close()


def run_output_space(dev, oracle):
    """a first buffer of 40 bytes (the header alone is 33), then 64-byte buffers; then
    a chain that ends too early"""
    rng = np.random.default_rng(4)
    usr = B.GlzUsr(buf_size=64)
    gd = B.GlzDictionary(dev, 1 << 16, 1, usr=usr)
    ge = B.GlzEncoder(dev, gd, 0, usr=usr)
    od = oracle.GlzDict(1 << 16, 1)
    try:
        px = pixels_for(T32, small_desktop(96, 30, 9), rng)
        want, _ = od.encode(T32, 96, 30, 384, [px])
        got, _ = ge.encode(T32, 96, 30, 384, [px])
        assert len(usr.out.bufs) > 10 and bytes(got) == bytes(want)

        left = [3]
        real_more = usr.out.more

        def stingy():
            left[0] -= 1
            return real_more() if left[0] > 0 else np.empty(0, dtype=np.uint8)

        usr.out.more = stingy
        noise = np.zeros((30, 96, 4), np.uint8)
        noise[..., :3] = rng.integers(0, 256, (30, 96, 3))
        noise = noise.reshape(30, 384)
        with pytest.raises(B.SpiceGpuError, match="no more bytes"):
            ge.encode(T32, 96, 30, 384, [noise])
        usr.out.more = real_more
        # the reference has added the image to the window by then (pre_encode ran), so has the oracle
        od.encode(T32, 96, 30, 384, [noise])
        nxt = pixels_for(T32, small_desktop(96, 30, 10), rng)
        want, _ = od.encode(T32, 96, 30, 384, [nxt])
        got, _ = ge.encode(T32, 96, 30, 384, [nxt])
        assert bytes(got) == bytes(want)
    finally:
        ge.close()
        gd.close()
        od.close()


def test_sim_limits(sim, oracle):
    run_limits(sim, oracle)


def test_sim_output_space(sim, oracle):
    run_output_space(sim, oracle)


@pytest.mark.gpu
def test_gpu_limits_and_output_space(gpu, oracle):
    run_limits(gpu, oracle)
    run_output_space(gpu, oracle)


@pytest.mark.gpu
@pytest.mark.parametrize("typ", [B.LZ_IMAGE_TYPE_RGB32, B.LZ_IMAGE_TYPE_RGBA, B.LZ_IMAGE_TYPE_RGB16])
def test_gpu_4k_frames(gpu, oracle, typ):
    """3840x2160 (8.3 Mpx, more than half the 16 Mi-pixel window: every image evicts
    its predecessor's predecessor) in three pixel formats"""
    from spice_amd import frames
    seq = frames.VideoSequence(3840, 2160, seed=2)
    rng = np.random.default_rng(1)
    pair = GlzPair(gpu, oracle, 16 * 1024 * 1024)
    try:
        for k in range(3):
            px = pixels_for(typ, seq.next(), rng)
            pair.encode(typ, 3840, 2160, px.shape[1], split_rows(px, 1 + k), what="4K frame %d" % k)
    finally:
        pair.close()
