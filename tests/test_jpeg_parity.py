"""MJPEG / still JPEG: the HIP kernels against libjpeg's own output (the golden
vectors) and against the CPU oracle, byte for byte, through the C ABI."""
import hashlib

import numpy as np
import pytest

from spice_amd import binding as B
from spice_amd import frames

from helpers import assert_same_stream, split_rows
from test_oracle_pinning import case_pixels, jpeg_cases


def batch_encode(dev, items):
    """items: (pixels2d, w, h, fmt, q, dct, stride_sign) -> list of JFIF byte arrays."""
    jobs, bufs = [], []
    for (px, w, h, fmt, q, dct, sign) in items:
        px = np.ascontiguousarray(px)
        stride = px.shape[1]
        d_px = dev.to_device(px)
        cap = dev.L.spice_gpu_jpeg_out_bound(w, h)
        d_out = dev.malloc(cap)
        first = d_px if sign > 0 else d_px + (h - 1) * stride
        jobs.append(dict(pixels=first, format=fmt, width=w, height=h, stride=sign * stride,
                         quality=q, dct=dct, out=d_out, out_capacity=cap))
        bufs.append((d_px, d_out))
    sizes = B.jpeg_encode_batch(dev, jobs)
    outs = [dev.download(b[1], n) for b, n in zip(bufs, sizes)]
    for d_px, d_out in bufs:
        dev.free(d_px)
        dev.free(d_out)
    return outs


def run_golden(dev, oracle, max_pixels, min_pixels=0, chunk=24):
    cases = [c for c in jpeg_cases(max_pixels) if c["w"] * c["h"] >= min_pixels]
    assert cases
    for i in range(0, len(cases), chunk):
        part = cases[i:i + chunk]
        items = [(case_pixels(oracle, c), c["w"], c["h"], c["fmt"], c["q"], c["dct"], 1) for c in part]
        for c, got in zip(part, batch_encode(dev, items)):
            assert len(got) == c["len"], c
            assert hashlib.sha256(got.tobytes()).hexdigest() == c["sha256"], c


def run_mjpeg_api(dev, oracle, sizes):
    """mjpeg_encoder_start_frame / encode_scanline / end_frame, frame after frame
    on one encoder (IFAST, quality 70, server/mjpeg_encoder.c:33,820)."""
    enc = B.MJpegEncoder(dev)
    try:
        for k, (w, h, fmt) in enumerate(sizes):
            img = frames.golden_image(w, h, "desktop" if k % 2 else "smooth", k)
            px = {B.BITMAP_FMT_32BIT: img.reshape(h, -1), B.BITMAP_FMT_RGBA: img.reshape(h, -1),
                  B.BITMAP_FMT_24BIT: frames.to_bgr24(img), B.BITMAP_FMT_16BIT: frames.to_rgb16(img)}[fmt]
            got = enc.encode_frame(fmt, w, h, px, px.shape[1])
            want = oracle.jpeg_encode(px, w, h, px.shape[1], fmt, 70, oracle.DCT_IFAST)
            assert_same_stream(want, got, "mjpeg frame %d" % k)
    finally:
        enc.close()


def run_still_api(dev, oracle):
    """jpeg_encode as red_jpeg_compress_image calls it (server/red_worker.c:6243-6262):
    quality 85, ISLOW, chunked rows through more_lines, bottom-up bitmaps walked
    from the last chunk with a negative stride, output through 64 KiB buffers."""
    enc = B.JpegEncoder(dev)
    try:
        for k, (w, h, nch, top_down) in enumerate([(320, 200, 1, True), (333, 77, 3, True),
                                                   (200, 150, 4, False), (64, 64, 2, False)]):
            img = frames.golden_image(w, h, "noise" if k == 1 else "desktop", 40 + k)
            px = img.reshape(h, -1)
            got = enc.encode(85, B.JPEG_IMAGE_TYPE_BGRX32, w, h, w * 4, split_rows(px, nch),
                             top_down=top_down)
            logical = px if top_down else px[::-1]
            want = oracle.jpeg_encode(np.ascontiguousarray(logical), w, h, w * 4,
                                      oracle.FMT_32BIT, 85, oracle.DCT_ISLOW)
            assert_same_stream(want, got, "still %d" % k)
    finally:
        enc.close()


def run_negative_stride_batch(dev, oracle):
    w, h = 100, 37
    px = frames.golden_image(w, h, "smooth", 3).reshape(h, -1)
    got = batch_encode(dev, [(px, w, h, B.BITMAP_FMT_32BIT, 80, 1, -1)])[0]
    want = oracle.jpeg_encode(np.ascontiguousarray(px[::-1]), w, h, w * 4, oracle.FMT_32BIT, 80, 1)
    assert_same_stream(want, got, "bottom-up batch")


def run_capacity_error(dev):
    w, h = 64, 64
    px = frames.golden_image(w, h, "noise", 1).reshape(h, -1)
    d_px, d_out = dev.to_device(px), dev.malloc(4096)
    with pytest.raises(B.SpiceGpuError):
        B.jpeg_encode_batch(dev, [dict(pixels=d_px, format=8, width=w, height=h, stride=w * 4,
                                       quality=100, dct=0, out=d_out, out_capacity=700)])
    dev.free(d_px)
    dev.free(d_out)


# ------------------------------------------------------------------ CPU (sim)
def test_sim_golden_small(sim, oracle):
    run_golden(sim, oracle, 64 * 48)


def test_sim_mjpeg_api(sim, oracle):
    run_mjpeg_api(sim, oracle, [(64, 48, 8), (33, 17, 7), (700, 20, 6), (17, 33, 9)])


def test_sim_still_api(sim, oracle):
    run_still_api(sim, oracle)


def test_sim_negative_stride_and_capacity(sim, oracle):
    run_negative_stride_batch(sim, oracle)
    run_capacity_error(sim)


def test_group_row_by_reciprocal_is_exact():
    """The strip kernel finds a workgroup's MCU row as (group x groups_rcp) >> 32 with
    groups_rcp = 2^32 / groups_per_row + 1 from the job record (jpeg_host.cpp build_job,
    jpeg_kernels.hip jpeg_strip_body), one group per row being tested for apart.  The product
    is the integer quotient while group x groups_per_row < 2^32; a JPEG is at most 65 535
    pixels a side = 4096 MCUs = 103 groups of 40 MCUs a row and 4096 MCU rows, and every group
    index of every such image is walked here."""
    for gpr in range(2, 104):
        rcp = (1 << 32) // gpr + 1
        assert rcp < (1 << 32)
        g = np.arange(gpr * 4096, dtype=np.uint64)
        assert np.array_equal((g * np.uint64(rcp)) >> np.uint64(32), g // np.uint64(gpr)), gpr


def test_quantiser_in_single_precision_is_exact():
    """the arithmetic claim behind the strip kernel's quantiser (jpeg_kernels.hip) and the tables
    jpeg_host.cpp fills: with rcp the single-precision number at or just above 1/d,
    trunc(float32(a) * rcp) = a // d for every dividend a = |v| + d/2 below 2^17 + d and every
    divisor a quantisation table can hold (jcdctmgr.c divides |v| + d/2 by d and restores the sign;
    the kernel does the same on v + copysign(d/2, v) and truncates towards zero)"""
    a = np.arange(0, (1 << 17) + 4200, dtype=np.uint32)
    af = a.astype(np.float32)
    for d in range(1, 4200):
        rcp = np.float32(1.0 / d)
        if float(rcp) * d < 1.0:
            rcp = np.nextafter(rcp, np.float32(2.0))
        assert float(rcp) * d >= 1.0 and float(np.nextafter(rcp, np.float32(0.0))) * d < 1.0 or float(rcp) * d == 1.0
        q = (af * rcp).astype(np.int64)          # float32 product, rounded to nearest, then truncated
        assert np.array_equal(q, a.astype(np.int64) // d), d
        qn = ((-af) * rcp).astype(np.int64)      # the negative side truncates towards zero as well
        assert np.array_equal(qn, -(a.astype(np.int64) // d)), d


def test_ifast_dct_in_single_precision_is_exact():
    """the claim behind jpeg_fdct_ifast_f2: every product of jfdctfst.c's passes stays below 2^24 in
    magnitude for 8-bit samples, so floor(float32(x) * float32(c / 256)) is x * c >> 8 (worst-case
    magnitudes of the operands of each multiplication, pass by pass)"""
    m = 255  # unsigned samples: the kernel removes the level shift from the DC term afterwards
    for _ in range(2):
        t = 2 * m                       # t0..t7 of a pass
        t1x = 2 * t                     # t10..t13 and t10..t12 of the odd part
        prods = [(2 * t1x, 181), (2 * t1x, 98), (t1x, 139), (t1x, 334), (t1x, 181)]
        for x, c in prods:
            assert x * c < 1 << 24, (x, c)
        z = max((x * c) >> 8 for x, c in prods)
        m = max(4 * t, t + 3 * z)       # a bound on what the pass hands to the next (d0 = sum of all eight; odd outputs)
    xs = np.arange(-(1 << 16), 1 << 16, dtype=np.int64)
    for c in (98, 139, 181, 334):
        want = (xs * c) >> 8
        got = np.floor(xs.astype(np.float32) * np.float32(c / 256.0)).astype(np.int64)
        keep = np.abs(xs * c) < (1 << 24)
        assert np.array_equal(want[keep], got[keep]), c


# ------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_gpu_golden_up_to_720p(gpu, oracle):
    run_golden(gpu, oracle, 1280 * 720)


@pytest.mark.gpu
def test_gpu_golden_4k_and_1080p(gpu, oracle):
    """BASELINE configs[2]: 3840x2160 frames, IFAST q70 (what the reference's
    stream encoder emits) and ISLOW q85, against libjpeg's output."""
    run_golden(gpu, oracle, 3840 * 2160, min_pixels=1280 * 720 + 1, chunk=3)


@pytest.mark.gpu
def test_gpu_every_colour(gpu, oracle):
    """all 2^24 colours, each filling a 2x2 block (so that the chroma averages are the colour's
    own Cb and Cr), at quality 100: the colour conversion has no input left untried"""
    w = h = 4096
    for part in range(4):
        c = np.arange(part << 22, (part + 1) << 22, dtype=np.uint32).reshape(h // 2, w // 2)
        px = np.repeat(np.repeat(c, 2, axis=0), 2, axis=1).view(np.uint8).reshape(h, w * 4)
        want = oracle.jpeg_encode(px, w, h, w * 4, oracle.FMT_32BIT, 100, oracle.DCT_ISLOW)
        got = batch_encode(gpu, [(px, w, h, B.BITMAP_FMT_32BIT, 100, 0, 1)])[0]
        assert_same_stream(want, got, "colours %d..%d" % (part << 22, ((part + 1) << 22) - 1))


@pytest.mark.gpu
def test_gpu_mjpeg_api(gpu, oracle):
    run_mjpeg_api(gpu, oracle, [(640, 480, 8), (333, 177, 7), (1280, 720, 6), (17, 33, 9), (1920, 1080, 8)])


@pytest.mark.gpu
def test_gpu_mjpeg_api_4k_golden(gpu):
    """BASELINE configs[2] through the reference's own call shape: a 3840x2160 BGRX frame handed to
    mjpeg_encoder_start_frame / encode_scanline x 2160 / end_frame (server/red_worker.c:8341-8377),
    IFAST quality 70 as server/mjpeg_encoder.c:820-822 sets it, against the SHA-256 of real
    libjpeg's output for that frame (tests/golden/jpeg_golden.json)."""
    cases = [c for c in jpeg_cases(3840 * 2160)
             if c["w"] == 3840 and c["h"] == 2160 and c["fmt"] == B.BITMAP_FMT_32BIT and c["q"] == 70
             and c["dct"] == B.DCT_IFAST]
    assert cases
    enc = B.MJpegEncoder(gpu)
    try:
        for c in cases[:2]:
            px = frames.golden_image(c["w"], c["h"], c["kind"], c["seed"]).reshape(c["h"], -1)
            got = enc.encode_frame(c["fmt"], c["w"], c["h"], px, px.shape[1])
            assert len(got) == c["len"], c
            assert hashlib.sha256(np.asarray(got).tobytes()).hexdigest() == c["sha256"], c
    finally:
        enc.close()


@pytest.mark.gpu
def test_gpu_still_api(gpu, oracle):
    run_still_api(gpu, oracle)


@pytest.mark.gpu
def test_gpu_negative_stride_and_capacity(gpu, oracle):
    run_negative_stride_batch(gpu, oracle)
    run_capacity_error(gpu)


@pytest.mark.gpu
def test_gpu_video_frames_match_oracle(gpu, oracle):
    seq = frames.VideoSequence(1920, 1080, seed=7)
    items = []
    for _ in range(3):
        f = seq.next().reshape(1080, -1)
        items.append((f, 1920, 1080, B.BITMAP_FMT_32BIT, 70, 1, 1))
        items.append((f, 1920, 1080, B.BITMAP_FMT_32BIT, 85, 0, 1))
    for it, got in zip(items, batch_encode(gpu, items)):
        want = oracle.jpeg_encode(it[0], it[1], it[2], it[0].shape[1], oracle.FMT_32BIT, it[4], it[5])
        assert_same_stream(want, got, "video frame")


def run_batch_plan(dev, oracle):
    """A prebuilt batch submitted twice with the frames swapped between the jobs."""
    w, h = 96, 48
    ims = [frames.golden_image(w, h, "desktop", s) for s in (1, 2)]
    px = [im.reshape(h, w * 4) for im in ims]
    want = [oracle.jpeg_encode(p, w, h, w * 4, 8, 70, B.DCT_IFAST) for p in px]
    d_px = [dev.to_device(p) for p in px]
    cap = dev.L.spice_gpu_jpeg_out_bound(w, h)
    d_out = [dev.malloc(cap) for _ in px]
    plan = B.JpegBatchPlan(dev, [dict(pixels=d_px[k], format=B.BITMAP_FMT_32BIT, width=w, height=h,
                                      stride=w * 4, quality=70, dct=B.DCT_IFAST, out=d_out[k],
                                      out_capacity=cap) for k in range(2)])
    for order in ((0, 1), (1, 0), (1, 1)):
        sizes = plan.run(np.array([d_px[k] for k in order], dtype=np.uint64))
        for j, k in enumerate(order):
            assert_same_stream(want[k], dev.download(d_out[j], int(sizes[j])), "order %s job %d" % (order, j))
    for p in d_px + d_out:
        dev.free(p)


def run_two_in_flight(dev, oracle, w=160, h=96, rounds=6):
    """Two batches in flight (submit, submit, finish, submit, finish ...): every batch's streams
    equal libjpeg's, whichever of the two buffer sets they land in; a third submit, a finish of
    the wrong batch and a synchronous batch in between are refused without breaking the pipeline."""
    kinds = ("desktop", "noise", "smooth")
    px = [frames.golden_image(w, h, kinds[s % 3], 11 + s).reshape(h, w * 4) for s in range(4)]
    want = {(k, q, d): oracle.jpeg_encode(px[k], w, h, w * 4, 8, q, d)
            for k in range(4) for (q, d) in ((70, B.DCT_IFAST), (85, B.DCT_ISLOW))}
    d_px = [dev.to_device(p) for p in px]
    cap = dev.L.spice_gpu_jpeg_out_bound(w, h)
    n = 3
    d_out = [dev.malloc(cap) for _ in range(n)]
    d_out2 = [dev.malloc(cap) for _ in range(n)]
    params = [(70, B.DCT_IFAST), (85, B.DCT_ISLOW), (70, B.DCT_IFAST)]
    plan = B.JpegBatchPlan(dev, [dict(pixels=d_px[0], format=B.BITMAP_FMT_32BIT, width=w, height=h, stride=w * 4,
                                      quality=params[j][0], dct=params[j][1], out=d_out[j], out_capacity=cap)
                                 for j in range(n)], alt_outs=d_out2)
    orders = [[(r + j) % 4 for j in range(n)] for r in range(rounds)]
    pending = []

    def check_oldest():
        sizes = plan.finish()
        order = pending.pop(0)
        outs = plan.last_outs or d_out
        for j, k in enumerate(order):
            assert_same_stream(want[(k,) + params[j]], dev.download(outs[j], int(sizes[j])), "pipelined job %d" % j)

    for r, order in enumerate(orders):
        if plan.in_flight() == 2:
            check_oldest()
        plan.submit(np.array([d_px[k] for k in order], dtype=np.uint64))
        pending.append(order)
        if r == 1:
            assert plan.in_flight() == 2
            with pytest.raises(B.SpiceGpuError):        # a third batch
                plan.submit(np.array([d_px[k] for k in order], dtype=np.uint64))
            with pytest.raises(B.SpiceGpuError):        # a synchronous batch while the thread has flights
                plan.run()
            wrong = (B.SpiceGpuJpegJob * (n + 1))()
            assert dev.L.spice_gpu_jpeg_finish(wrong, n + 1) != 0   # not the oldest batch's job count
    while plan.in_flight():
        check_oldest()
    # the synchronous way still works afterwards (the gate was released)
    sizes = plan.run(np.array([d_px[k] for k in (3, 2, 1)], dtype=np.uint64))
    for j, k in enumerate((3, 2, 1)):
        assert_same_stream(want[(k,) + params[j]], dev.download(d_out[j], int(sizes[j])), "after the pipeline")
    for p in d_px + d_out + d_out2:
        dev.free(p)


def test_sim_batch_plan(sim, oracle):
    run_batch_plan(sim, oracle)


def test_sim_two_batches_in_flight(sim, oracle):
    run_two_in_flight(sim, oracle)


@pytest.mark.gpu
def test_gpu_two_batches_in_flight(gpu, oracle):
    run_two_in_flight(gpu, oracle, 1280, 720, rounds=9)


@pytest.mark.gpu
def test_gpu_batch_plan(gpu, oracle):
    run_batch_plan(gpu, oracle)
