"""Threads side by side in one process, as a RedWorker thread per display would call the
library: a thread that keeps JPEG batches in flight (submit, submit, finish, submit ...)
must not keep other threads out -- they create dictionaries, run the synchronous GLZ and
JPEG drop-ins and their own batches meanwhile.  Nothing is held from one API call to the
next except a batch's own flight slot and, for GLZ, its dictionaries' locks."""
import threading

import numpy as np
import pytest

from spice_amd import binding as B
from spice_amd import frames
from helpers import assert_same_stream


def run_pipeline_beside_other_threads(dev, oracle, w=96, h=48, rounds=6):
    px = [frames.golden_image(w, h, ("desktop", "noise", "smooth")[k % 3], 21 + k).reshape(h, w * 4)
          for k in range(3)]
    want_jpeg = [oracle.jpeg_encode(p, w, h, w * 4, 8, 70, B.DCT_IFAST) for p in px]
    d_px = [dev.to_device(p) for p in px]
    cap = dev.L.spice_gpu_jpeg_out_bound(w, h)
    outs = [dev.malloc(cap) for _ in range(3)]
    outs2 = [dev.malloc(cap) for _ in range(3)]
    plan = B.JpegBatchPlan(dev, [dict(pixels=d_px[k], format=B.BITMAP_FMT_32BIT, width=w, height=h,
                                      stride=w * 4, quality=70, dct=B.DCT_IFAST, out=outs[k],
                                      out_capacity=cap) for k in range(3)], alt_outs=outs2)
    errors = []
    in_flight = threading.Event()     # thread A has two batches in flight
    b_done = threading.Event()

    def thread_b():
        """everything another worker thread may want while A's batches are in flight"""
        try:
            assert in_flight.wait(60)
            seq = frames.DesktopSequence(w, h, seed=77)
            od = oracle.GlzDict(1 << 20, 1)
            gd = B.GlzDictionary(dev, 1 << 20, 1)           # glz_enc_dictionary_create -> reset
            ge = B.GlzEncoder(dev, gd, 0)
            for f in range(3):
                img = seq.next().reshape(h, -1)
                want, _ = od.encode(B.LZ_IMAGE_TYPE_RGB32, w, h, w * 4, [img])
                got, _ = ge.encode(B.LZ_IMAGE_TYPE_RGB32, w, h, w * 4, [img])     # synchronous glz_encode
                assert_same_stream(want, got, "glz_encode beside a JPEG pipeline, frame %d" % f)
            gd.reset()
            # the synchronous still-JPEG drop-in (its own flight on the library's stream)
            je = B.JpegEncoder(dev)
            got = je.encode(85, B.JPEG_IMAGE_TYPE_BGRX32, w, h, w * 4, [px[0]])
            assert_same_stream(oracle.jpeg_encode(px[0], w, h, w * 4, 8, 85, B.DCT_ISLOW), got, "jpeg_encode")
            je.close()
            b_done.set()
            # a batch of this thread's own needs one of the two flight slots: it waits until
            # thread A finishes a batch (A is told to go on above)
            d_img = dev.to_device(px[1])
            out = dev.malloc(cap)
            n = B.jpeg_encode_batch(dev, [dict(pixels=d_img, format=B.BITMAP_FMT_32BIT, width=w, height=h,
                                               stride=w * 4, quality=70, dct=B.DCT_IFAST, out=out,
                                               out_capacity=cap)])[0]
            assert_same_stream(want_jpeg[1], dev.download(out, n), "another thread's batch")
            dev.free(d_img)
            dev.free(out)
            ge.close()
            gd.close()
            od.close()
        except BaseException as e:       # noqa: BLE001 -- handed to the main thread
            errors.append(e)
        finally:
            b_done.set()

    tb = threading.Thread(target=thread_b)
    tb.start()
    try:
        plan.submit()
        plan.submit()
        in_flight.set()
        # A does not finish anything until B is through: were the library still held between
        # A's calls, B would wait for ever here
        assert b_done.wait(120), "a thread was kept out of the library by another thread's batches in flight"
        for r in range(rounds):
            sizes = plan.finish()
            o = plan.last_outs or outs
            for k in range(3):
                assert_same_stream(want_jpeg[k], dev.download(o[k], int(sizes[k])), "pipelined frame %d" % k)
            plan.submit()
        while plan.in_flight():
            plan.finish()
    finally:
        in_flight.set()
        tb.join()
        for p in d_px + outs + outs2:
            dev.free(p)
    if errors:
        raise errors[0]


def test_sim_pipeline_beside_other_threads(sim, oracle):
    run_pipeline_beside_other_threads(sim, oracle)


@pytest.mark.gpu
def test_gpu_pipeline_beside_other_threads(gpu, oracle):
    run_pipeline_beside_other_threads(gpu, oracle, 640, 360)
