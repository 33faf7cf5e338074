"""BASELINE configs[3] and configs[4] through the batch interface, and the client's
view of what the GPU produces.

multimon    one client, four display channels, four encoders on ONE dictionary
            (server/red_worker.c:9677-9754): per tick four GLZ updates (70 % of a
            monitor) in channel order and four MJPEG frames (the other 30 %).
sessions64  the 32 GLZ sessions of 64 concurrent ones, one batch per frame.  The
            other 32 are the sessions the reference codes with QUIC
            (spice-common, not in the reference tree): not built, not tested.

Every GLZ stream the library produces is compared with the CPU oracle's AND
decoded by the restated client decoder (client/glz_decode_tmpl.c:179-319 with the
window of client/glz_decoder_window.cpp): the decoder drops images older than the
stream's own window head, so a reference to an image the encoder should have
evicted, or a stream that depends on bookkeeping gone wrong, fails to decode; the
decoded pixels must be the input's.  After every batch the window's bookkeeping
is compared with an independent model (window_model.py)."""
import numpy as np
import pytest

from spice_amd import binding as B
from spice_amd import frames

from helpers import assert_same_stream, pixels_for, small_desktop, split_rows
from window_model import WindowModel

T32 = B.LZ_IMAGE_TYPE_RGB32


def assert_decodes_to(dec, stream, px, typ, w, h, what):
    """px: the encoder's input rows (h, stride)"""
    stride = px.shape[1]
    out = dec.decode(stream, h * stride)
    assert out["used"] == len(stream) and (out["type"], out["width"], out["height"]) == (typ, w, h), what
    got = out["pixels"][:h * stride].reshape(h, stride)
    want = px
    if typ == T32:                       # the pad byte is not transmitted
        want = px.reshape(h, w, 4).copy()
        want[..., 3] = 0
        want = want.reshape(h, -1)
    if typ == B.LZ_IMAGE_TYPE_RGB16:     # bit 15 takes no part in matches
        assert ((got.view(np.uint16) ^ want.view(np.uint16)) & 0x7fff == 0).all(), what
    else:
        assert (got == want).all(), what + ": decoded pixels differ from the input"


def run_multimon(dev, oracle, w, h, ticks, window, monitors=4):
    client = frames.MultimonClient(w, h, monitors, seed=77)
    gr, vr = client.glz_rows, client.video_rows
    od = oracle.GlzDict(window, monitors)
    gd = B.GlzDictionary(dev, window, monitors)
    ges = [B.GlzEncoder(dev, gd, m) for m in range(monitors)]
    model = WindowModel(window, monitors)
    dec = oracle.GlzDecoder()
    gcap = dev.L.spice_gpu_glz_out_bound(T32, w, gr)
    vcap = max(4096, w * vr)
    gouts = [dev.malloc(gcap) for _ in range(monitors)]
    vouts = [dev.malloc(vcap) for _ in range(monitors)]
    tag = 0
    try:
        for t in range(ticks):
            tick = client.tick()
            gpx = [np.ascontiguousarray(g).reshape(gr, -1) for g, _ in tick]
            vpx = [np.ascontiguousarray(v).reshape(vr, -1) for _, v in tick]
            held = [dev.to_device(p) for p in gpx + vpx]
            try:
                sizes, _ = B.glz_encode_batch(dev, [
                    dict(encoder=ges[m], type=T32, width=w, height=gr, stride=w * 4, chunks=[held[m]],
                         chunk_rows=[gr], out=gouts[m], out_capacity=gcap, usr_context=9000 + tag + m)
                    for m in range(monitors)], split=bool(t % 2))
                vsizes = B.jpeg_encode_batch(dev, [
                    dict(pixels=held[monitors + m], format=B.BITMAP_FMT_32BIT, width=w, height=vr,
                         stride=w * 4, quality=70, dct=B.DCT_IFAST, out=vouts[m], out_capacity=vcap)
                    for m in range(monitors)])
                for m in range(monitors):
                    what = "tick %d channel %d" % (t, m)
                    want, _ = od.encode(T32, w, gr, w * 4, [gpx[m]], encoder_id=m, usr_tag=9000 + tag + m)
                    got = dev.download(gouts[m], sizes[m])
                    assert_same_stream(want, got, what)
                    assert_decodes_to(dec, got, gpx[m], T32, w, gr, what)
                    model.add(T32, gr, w * 4, [gr], 9000 + tag + m)
                    wantv = oracle.jpeg_encode(vpx[m], w, vr, w * 4, oracle.FMT_32BIT, 70, oracle.DCT_IFAST)
                    assert_same_stream(wantv, dev.download(vouts[m], vsizes[m]), what + " video")
                tag += monitors
                assert od.pop_freed() == model.pop_freed() == gd.usr.freed_images
                gd.usr.freed_images.clear()
                model.check(gd.dump(), "tick %d" % t)
            finally:
                for p in held:
                    dev.free(p)
    finally:
        for p in gouts + vouts:
            dev.free(p)
        for e in ges:
            e.close()
        gd.close()
        od.close()
        dec.close()


def run_sessions64(dev, oracle, w, h, nframes, window):
    gen = frames.Sessions64(w, h)
    ids = gen.session_ids("glz")
    assert len(ids) == 32 and len(gen.session_ids("quic")) == 32
    seqs = [gen.sequence(c) for c in ids]
    ods = [oracle.GlzDict(window, 1) for _ in ids]
    gds = [B.GlzDictionary(dev, window, 1) for _ in ids]
    ges = [B.GlzEncoder(dev, d, 0) for d in gds]
    decs = [oracle.GlzDecoder() for _ in ids]
    cap = dev.L.spice_gpu_glz_out_bound(T32, w, h)
    outs = [dev.malloc(cap) for _ in ids]
    try:
        for f in range(nframes):
            px = [np.ascontiguousarray(s.next()).reshape(h, -1) for s in seqs]
            held = [dev.to_device(p) for p in px]
            try:
                sizes, _ = B.glz_encode_batch(dev, [
                    dict(encoder=ges[k], type=T32, width=w, height=h, stride=w * 4, chunks=[held[k]],
                         chunk_rows=[h], out=outs[k], out_capacity=cap) for k in range(len(ids))])
                for k, c in enumerate(ids):
                    what = "frame %d session %d" % (f, c)
                    want, _ = ods[k].encode(T32, w, h, w * 4, [px[k]])
                    got = dev.download(outs[k], sizes[k])
                    assert_same_stream(want, got, what)
                    if k % 8 == 0:
                        assert_decodes_to(decs[k], got, px[k], T32, w, h, what)
            finally:
                for p in held:
                    dev.free(p)
    finally:
        for k in range(len(ids)):
            dev.free(outs[k])
            ges[k].close()
            gds[k].close()
            ods[k].close()
            decs[k].close()


def run_decode_of_device_streams(dev, oracle):
    """all four pixel types, ragged sizes, several chunks, both row orders, a window small
    enough to evict every few images, an image removed, two encoders taking turns: what the
    library emits must decode to the pixels that went in"""
    rng = np.random.default_rng(12)
    sizes = [(64, 32), (100, 60), (7, 3), (200, 50), (130, 70), (33, 17), (1, 1), (96, 40)]
    for typ in (T32, B.LZ_IMAGE_TYPE_RGBA, B.LZ_IMAGE_TYPE_RGB24, B.LZ_IMAGE_TYPE_RGB16):
        window = 26000
        gd = B.GlzDictionary(dev, window, 2)
        enc = [B.GlzEncoder(dev, gd, 0), B.GlzEncoder(dev, gd, 1)]
        model = WindowModel(window, 2)
        dec = oracle.GlzDecoder()
        try:
            for i in range(40):
                w, h = sizes[i % len(sizes)]
                px = pixels_for(typ, small_desktop(w, h, i // 3), rng)
                chunks = split_rows(px, 1 + i % 3)
                top_down = bool(i % 2)
                s, image = enc[(i // 2) % 2].encode(typ, w, h, px.shape[1], chunks, top_down=top_down,
                                                    usr_context=100 + i)
                what = "type %d image %d (%dx%d)" % (typ, i, w, h)
                assert_decodes_to(dec, s, px, typ, w, h, what)
                model.add(typ, h, px.shape[1], [c.shape[0] for c in chunks], 100 + i)
                assert model.pop_freed() == gd.usr.freed_images, what
                gd.usr.freed_images.clear()
                if i % 11 == 5:
                    gd.remove_image(image)
                    model.remove(100 + i)
                model.check(gd.dump(), what)
        finally:
            for e in enc:
                e.close()
            gd.close()
            dec.close()


def test_sim_multimon(sim, oracle):
    # the window holds three of the 320x126 desktop regions: every tick evicts
    run_multimon(sim, oracle, 320, 180, ticks=3, window=320 * 126 * 3 + 50)


def test_sim_sessions64(sim, oracle):
    run_sessions64(sim, oracle, 96, 54, nframes=2, window=1 << 14)


def test_sim_decode_of_device_streams(sim, oracle):
    run_decode_of_device_streams(sim, oracle)


@pytest.mark.gpu
@pytest.mark.usefixtures("both_glz_kernels")     # both GLZ parse kernels (tests/conftest.py)
def test_gpu_multimon_4k(gpu, oracle):
    """four 3840x2160 monitors on one 16 Mi-pixel dictionary: a 3840x1512 region is 5.8 M
    pixels, so the window holds two and every image evicts its predecessor's predecessor"""
    run_multimon(gpu, oracle, 3840, 2160, ticks=3, window=16 * 1024 * 1024)


@pytest.mark.gpu
@pytest.mark.usefixtures("both_glz_kernels")     # both GLZ parse kernels (tests/conftest.py)
def test_gpu_sessions64_glz_half(gpu, oracle):
    run_sessions64(gpu, oracle, 1920, 1080, nframes=3, window=16 * 1024 * 1024)


@pytest.mark.gpu
@pytest.mark.usefixtures("both_glz_kernels")     # both GLZ parse kernels (tests/conftest.py)
def test_gpu_decode_of_device_streams(gpu, oracle):
    run_decode_of_device_streams(gpu, oracle)
