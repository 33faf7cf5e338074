"""The GLZ dictionary's life cycle through the drop-in API, against the oracle:
reset, migration restore (glz_enc_dictionary_get_restore_data / _restore,
server/glz_encoder_dictionary.c:96-160), and two encoders of one client sharing
a dictionary (server/red_worker.c:9677-9754; each encoder holds back the window
head while it encodes, glz_encoder_dictionary.c:550-633)."""
import numpy as np
import pytest

from spice_amd import binding as B

from helpers import GlzPair, assert_same_stream, pixels_for, small_desktop, split_rows

T32 = B.LZ_IMAGE_TYPE_RGB32


def images(n, seed0, w=120, h=48):
    rng = np.random.default_rng(seed0)
    return [pixels_for(T32, small_desktop(w, h, seed0 + k // 2), rng) for k in range(n)], w, h


def run_reset(dev, oracle):
    imgs, w, h = images(6, 40)
    od = oracle.GlzDict(20000, 1)
    gd = B.GlzDictionary(dev, 20000, 1)
    ge = B.GlzEncoder(dev, gd, 0)
    try:
        for k, px in enumerate(imgs):
            if k == 3:
                od.reset()
                gd.reset()
            so, _ = od.encode(T32, w, h, w * 4, [px], usr_tag=k + 1)
            sg, _ = ge.encode(T32, w, h, w * 4, [px], usr_context=k + 1)
            assert_same_stream(so, sg, "image %d around a reset" % k)
        # the reset freed every image that was still in the window
        assert od.pop_freed() == gd.usr.freed_images
    finally:
        ge.close()
        gd.close()
        od.close()


def run_restore(dev, oracle):
    imgs, w, h = images(6, 60)
    od = oracle.GlzDict(30000, 2)
    gd = B.GlzDictionary(dev, 30000, 2)
    ge = B.GlzEncoder(dev, gd, 0)
    for k, px in enumerate(imgs[:3]):
        so, _ = od.encode(T32, w, h, w * 4, [px], usr_tag=k + 1)
        sg, _ = ge.encode(T32, w, h, w * 4, [px], usr_context=k + 1)
        assert_same_stream(so, sg, "image %d before migration" % k)
    want = od.restore_data()
    got = gd.get_restore_data()
    assert (got.size, got.max_encoders, got.last_image_id) == want
    ge.close()
    gd.close()
    od.close()
    # the target host starts from the restore data: empty window, image ids go on
    od = oracle.GlzDict(want[0], want[1], restore_last_id=want[2])
    gd = B.GlzDictionary(dev, 0, 0, restore=got)
    ge = B.GlzEncoder(dev, gd, 1)
    try:
        assert gd.get_size() == want[0]
        for k, px in enumerate(imgs[3:]):
            so, _ = od.encode(T32, w, h, w * 4, [px], encoder_id=1, usr_tag=10 + k)
            sg, _ = ge.encode(T32, w, h, w * 4, [px], usr_context=10 + k)
            assert_same_stream(so, sg, "image %d after migration" % k)
    finally:
        ge.close()
        gd.close()
        od.close()


def run_two_encoders(dev, oracle):
    """Two display channels of one client take turns on one dictionary; the
    window is small, so images of both are evicted along the way."""
    imgs, w, h = images(10, 80)
    od = oracle.GlzDict(25000, 2)
    gd = B.GlzDictionary(dev, 25000, 2)
    enc = [B.GlzEncoder(dev, gd, 0), B.GlzEncoder(dev, gd, 1)]
    try:
        for k, px in enumerate(imgs):
            who = (k // 2) % 2
            chunks = split_rows(px, 1 + k % 2)
            so, _ = od.encode(T32, w, h, w * 4, chunks, encoder_id=who, usr_tag=k + 1)
            sg, _ = enc[who].encode(T32, w, h, w * 4, chunks, usr_context=k + 1)
            assert_same_stream(so, sg, "image %d by encoder %d" % (k, who))
            assert od.pop_freed() == gd.usr.freed_images
            gd.usr.freed_images.clear()
    finally:
        for e in enc:
            e.close()
        gd.close()
        od.close()


def run_restore_large_ids(dev, oracle):
    """Image ids at and beyond 2^24 and 2^32.  The header carries the id as two 32-bit words and
    the reference masks the low one with 0xffffff (encode_64, server/glz_encoder.c:83-87), so ids
    from 2^24 on are written wrongly -- and must be written wrongly in the same way: bytes 21..28
    of every stream are compared explicitly, the whole stream with the oracle's.  The ids come from
    glz_enc_dictionary_restore (migration, server/glz_encoder_dictionary.c:178-201), the only way
    a dictionary starts anywhere but at 0."""
    imgs, w, h = images(4, 90)
    for first_id in ((1 << 24) - 2, (1 << 32) - 2):
        od = oracle.GlzDict(30000, 1, restore_last_id=first_id)
        data = B.GlzEncDictRestoreData(30000, 1, first_id)
        gd = B.GlzDictionary(dev, 0, 0, restore=data)
        ge = B.GlzEncoder(dev, gd, 0)
        try:
            for k, px in enumerate(imgs):
                so, _ = od.encode(T32, w, h, w * 4, [px], usr_tag=k + 1)
                sg, _ = ge.encode(T32, w, h, w * 4, [px], usr_context=k + 1)
                image_id = first_id + k
                want_id = (image_id >> 32).to_bytes(4, "big") + (image_id & 0xffffff).to_bytes(4, "big")
                assert bytes(sg[21:29]) == want_id, "image id %#x in the header" % image_id
                assert_same_stream(so, sg, "image id %#x" % image_id)
            assert gd.get_restore_data().last_image_id == first_id + len(imgs)
        finally:
            ge.close()
            gd.close()
            od.close()


# do_match's reject of image distances above 2^24 - 1 (server/glz_encode_tmpl.c:226-229,
# glz_encoder.c:195-198) has no test: ids are handed out one by one, so a reference that far
# back needs more than 2^24 images in one window, a window holds at most 2^25 pixels
# (glz_encoder_dictionary.c:50), and red_compress_image sends nothing under 54 bytes -- 14 pixels
# of RGB32 -- through a codec (server/red_worker.c:6423-6426): at most 2.4 million images fit.
# The kernel carries the test all the same (glz_kernels.hip, scan phase: idist <= 16777215).
CASES = [run_reset, run_restore, run_two_encoders, run_restore_large_ids]


@pytest.mark.parametrize("case", CASES, ids=lambda f: f.__name__)
def test_sim_dictionary(sim, oracle, case):
    case(sim, oracle)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=lambda f: f.__name__)
def test_gpu_dictionary(gpu, oracle, case):
    case(gpu, oracle)


def run_arena_regrow(dev, oracle):
    """images of growing size through a small window: the window's pixel copies
    live in a FIFO ring of one window's bytes, which is re-made larger (live images
    moved) once the wasted run at its wrap point leaves no room.  Streams must
    not notice."""
    typ = B.LZ_IMAGE_TYPE_RGB32
    window = 200000
    pair = GlzPair(dev, oracle, window)
    rng = np.random.default_rng(8)
    try:
        before = pair.gd.info()
        assert before["arena_bytes"] == window * 4 + 65536 and before["arena_regrows"] == 0
        # the sixth image finds 132 KB + 272 KB free, in two pieces, and needs 396 KB in one
        heights = [200, 150, 280, 150, 330, 330, 330, 320, 250, 150, 330, 100]
        for k, h in enumerate(heights):
            flat = np.zeros((h, 300, 4), np.uint8)
            flat[..., :3] = small_desktop(300, 96, k // 3)[(np.arange(h) * 7) % 96][..., :3]
            px = pixels_for(typ, flat, rng)
            pair.encode(typ, 300, h, 1200, split_rows(px, 1 + k % 3), what="regrow %d" % k)
            info = pair.gd.info()
            assert info["window_pixels"] <= window and info["images"] >= 1
            assert info["arena_regrows"] == (0 if k < 5 else 1), k
        after = pair.gd.info()
        assert before["arena_bytes"] < after["arena_bytes"] <= 2 * window * 4 + 65536
    finally:
        pair.close()


def test_sim_arena_regrow(sim, oracle):
    run_arena_regrow(sim, oracle)


@pytest.mark.gpu
def test_gpu_arena_regrow(gpu, oracle):
    run_arena_regrow(gpu, oracle)


def run_worker_threads(dev, oracle, nthreads=4, nimages=6):
    """RedWorker threads, one per display, each with its own dictionary and encoder, call
    glz_encode at the same time (one worker thread per QXL device,
    server/red_dispatcher.c:1148): every stream must be the one the oracle makes for that
    display alone."""
    import threading
    # displays of unequal size: the calls that arrive together share a launch, several such
    # launches run at a time, and a small image must come out right beside a large one
    sizes = [((160, 64), (320, 120), (96, 48), (200, 90))[t % 4] for t in range(nthreads)]
    per = [images(nimages, 200 + 10 * t, *sizes[t])[0] for t in range(nthreads)]
    gds = [B.GlzDictionary(dev, 120000, 1) for _ in range(nthreads)]
    ges = [B.GlzEncoder(dev, d, 0) for d in gds]
    got = [[] for _ in range(nthreads)]
    errs = []

    def work(t):
        w, h = sizes[t]
        try:
            for px in per[t]:
                got[t].append(ges[t].encode(T32, w, h, w * 4, [px])[0])
        except Exception as e:              # noqa: BLE001 - reported below
            errs.append((t, e))

    ths = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    try:
        assert not errs, errs
        for t in range(nthreads):
            w, h = sizes[t]
            od = oracle.GlzDict(120000, 1)
            for k, px in enumerate(per[t]):
                want, _ = od.encode(T32, w, h, w * 4, [px])
                assert_same_stream(want, got[t][k], "thread %d image %d" % (t, k))
            od.close()
    finally:
        for e in ges:
            e.close()
        for d in gds:
            d.close()


def test_sim_worker_threads(sim, oracle):
    run_worker_threads(sim, oracle)


@pytest.mark.gpu
def test_gpu_worker_threads(gpu, oracle):
    run_worker_threads(gpu, oracle, nthreads=8, nimages=10)
