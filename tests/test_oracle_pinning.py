"""The oracle itself, pinned (CPU only).

JPEG: the restatement reproduces vectors that the real libjpeg produced
(tests/golden/jpeg_golden.json, made by tests/golden/make_jpeg_golden.py), and,
where oracle/_ref/libjpegref.so exists, fresh output of the real library.
GLZ: the reference cannot be built here, so the encoder's choices are unpinned;
the header layout is checked against bytes recorded from a reference run and
every stream is decoded by the restated client decoder.
"""
import base64
import hashlib
import json
import os

import numpy as np
import pytest

from spice_amd import frames

from helpers import pixels_for, small_desktop, split_rows

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def jpeg_cases(max_pixels):
    with open(os.path.join(GOLDEN, "jpeg_golden.json")) as f:
        return [c for c in json.load(f)["cases"] if c["w"] * c["h"] <= max_pixels]


def case_pixels(O, c):
    img = frames.golden_image(c["w"], c["h"], c["kind"], c["seed"])
    if c["fmt"] == O.FMT_32BIT:
        return img.reshape(c["h"], -1)
    if c["fmt"] == O.FMT_24BIT:
        return frames.to_bgr24(img)
    return frames.to_rgb16(img)


def test_jpeg_oracle_matches_libjpeg_golden(oracle):
    O = oracle
    cases = jpeg_cases(1280 * 720)
    assert len(cases) > 250
    for c in cases:
        px = case_pixels(O, c)
        got = O.jpeg_encode(px, c["w"], c["h"], px.shape[1], c["fmt"], c["q"], c["dct"])
        assert len(got) == c["len"], c
        assert hashlib.sha256(got.tobytes()).hexdigest() == c["sha256"], c
        if "bytes" in c:
            assert got.tobytes() == base64.b64decode(c["bytes"])


def test_jpeg_oracle_matches_live_libjpeg(oracle):
    O = oracle
    if O.ref() is None:
        pytest.skip("oracle/_ref/libjpegref.so not built on this machine")
    rng = np.random.default_rng(4)
    for (w, h) in [(16, 16), (33, 17), (100, 75), (1, 1), (17, 1), (250, 130)]:
        a = rng.integers(0, 256, (h, w * 4), dtype=np.uint8)
        for q in (20, 70, 85):
            for dct in (O.DCT_ISLOW, O.DCT_IFAST):
                want = O.ref_jpeg_encode(a, w, h, w * 4, O.FMT_32BIT, q, dct, 0)
                got = O.jpeg_encode(a, w, h, w * 4, O.FMT_32BIT, q, dct)
                assert len(want) == len(got) and (want == got).all(), (w, h, q, dct)


def test_jpeg_oracle_matches_live_libjpeg_at_the_extremes(oracle):
    """full-swing patterns at every spatial frequency, qualities 1..100: where the DCT's operands
    and the quantiser's dividends reach their bounds (the kernels' single-precision arithmetic is
    checked against the restatement on exactly these frames, tests/test_jpeg_fuzz.py)"""
    O = oracle
    if O.ref() is None:
        pytest.skip("oracle/_ref/libjpegref.so not built on this machine")
    from test_jpeg_fuzz import extreme_frames
    for name, im in extreme_frames(48, 32):
        a = im.reshape(32, 48 * 4)
        for q in (1, 2, 50, 99, 100):
            for dct in (O.DCT_ISLOW, O.DCT_IFAST):
                want = O.ref_jpeg_encode(a, 48, 32, 48 * 4, O.FMT_32BIT, q, dct, 0)
                got = O.jpeg_encode(a, 48, 32, 48 * 4, O.FMT_32BIT, q, dct)
                assert len(want) == len(got) and (want == got).all(), (name, q, dct)


def test_jpeg_colour_constants_of_v6b_and_ijg9_agree():
    """libjpeg 9 rounds two Cb coefficients differently from v6b / libjpeg-turbo
    (11058/21710 vs 11059/21709); over 8-bit input the results never differ."""
    r, g, b = np.meshgrid(np.arange(256), np.arange(256), np.arange(256), indexing="ij")
    off = (128 << 16) + 32767
    v6b = (-11059 * r - 21709 * g + 32768 * b + off) >> 16
    ijg9 = (-11058 * r - 21710 * g + 32768 * b + off) >> 16
    assert (v6b == ijg9).all()


def test_glz_header_matches_recorded_reference_bytes(oracle):
    O = oracle
    with open(os.path.join(GOLDEN, "glz_header.json")) as f:
        g = json.load(f)
    d = O.GlzDict(1 << 20, 1)
    img = small_desktop(g["width"], g["height"], 0).reshape(g["height"], -1)
    for _ in range(g["images_before"]):
        d.encode(g["type"], g["width"], g["height"], g["stride"], [img])
    s, _ = d.encode(g["type"], g["width"], g["height"], g["stride"], [img], top_down=bool(g["top_down"]))
    assert s[:33].tobytes().hex() == g["header_hex"]


@pytest.mark.parametrize("typ", [5, 6, 7, 8, 9])
def test_glz_oracle_round_trips_through_restated_decoder(oracle, typ):
    O = oracle
    rng = np.random.default_rng(typ)
    d = O.GlzDict(40000, 1)
    dec = O.GlzDecoder()
    sizes = [(64, 32), (100, 60), (7, 3), (1, 1), (3, 1), (200, 50), (2, 2), (130, 70)]
    for i in range(32):
        w, h = sizes[i % len(sizes)]
        img = small_desktop(w, h, i // 2)
        if typ == 5:
            px = np.ascontiguousarray(img[..., 0])
        else:
            px = pixels_for(typ, img, rng)
        stride = px.shape[1]
        s, handle = d.encode(typ, w, h, stride, split_rows(px, 1 if i % 3 else 3),
                             top_down=bool(i % 2))
        out = dec.decode(s, h * stride)
        assert out["used"] == len(s)
        assert (out["type"], out["width"], out["height"]) == (typ, w, h)
        got = out["pixels"][:h * stride].reshape(h, stride)
        want = px.copy()
        if typ == 8:   # the pad byte is not transmitted
            want = want.reshape(h, w, 4).copy()
            want[..., 3] = 0
            want = want.reshape(h, -1)
        if typ == 6:   # bit 15 is ignored by matches
            assert ((got.view(np.uint16) ^ want.view(np.uint16)) & 0x7fff == 0).all()
        else:
            assert (got == want).all()
        if i == 5:
            d.remove_image(handle)
    assert d.stats()["matches"] > 0 and d.stats()["rle_matches"] > 0
