"""The device-resident batch interface (spice_gpu_glz_encode_batch): many
sessions, one image each per call, pixels already in device memory and streams
left there.  Every session's stream is compared with the CPU oracle's, frame
after frame.  The library starts a batch's workgroups in the order of expected
cost, so the cases mix cheap and expensive sessions and check that results
still come back in the caller's job order."""
import numpy as np
import pytest

from spice_amd import binding as B
from spice_amd import frames

from helpers import assert_same_stream, pixels_for, small_desktop, split_rows
from window_model import WindowModel


class Session:
    def __init__(self, dev, oracle, typ, w, h, kind, seed, window, nchunks=1):
        self.dev, self.typ, self.w, self.h, self.kind, self.nchunks = dev, typ, w, h, kind, nchunks
        self.rng = np.random.default_rng(seed)
        self.seq = frames.DesktopSequence(max(w, 96), max(h, 96), seed)
        self.od = oracle.GlzDict(window, 1)
        self.gd = B.GlzDictionary(dev, window, 1)
        self.ge = B.GlzEncoder(dev, self.gd, 0)
        self.cap = dev.L.spice_gpu_glz_out_bound(typ, w, h)
        self.out = dev.malloc(self.cap)
        self.held = []

    def next_image(self):
        if self.kind == "noise":
            im = np.zeros((self.h, self.w, 4), np.uint8)
            im[..., :3] = self.rng.integers(0, 256, (self.h, self.w, 3))
        elif self.kind == "repeat":       # the same picture every time: one match to the segment's end
            im = frames.DesktopSequence(max(self.w, 96), max(self.h, 96), 77).next()[:self.h, :self.w].copy()
        elif self.kind == "flat":
            im = np.zeros((self.h, self.w, 4), np.uint8)
            im[..., :3] = self.rng.integers(0, 256, 3)
        else:
            im = self.seq.next()[:self.h, :self.w].copy()
        px = pixels_for(self.typ, im, self.rng)
        self.stride = px.shape[1]
        self.chunks = split_rows(px, self.nchunks)
        self.held = [self.dev.to_device(c) for c in self.chunks]
        return dict(encoder=self.ge, type=self.typ, width=self.w, height=self.h, stride=self.stride,
                    chunks=self.held, chunk_rows=[c.shape[0] for c in self.chunks], out=self.out,
                    out_capacity=self.cap)

    def check(self, size, what):
        want, _ = self.od.encode(self.typ, self.w, self.h, self.stride, self.chunks)
        got = self.dev.download(self.out, size)
        assert_same_stream(want, got, what)
        for p in self.held:
            self.dev.free(p)

    def close(self):
        self.dev.free(self.out)
        self.ge.close()
        self.gd.close()
        self.od.close()


def run_batches(dev, oracle, specs, nframes, window):
    sessions = [Session(dev, oracle, *spec, seed=31 + k, window=window) for k, spec in enumerate(specs)]
    try:
        for f in range(nframes):
            items = [s.next_image() for s in sessions]
            sizes, _ = B.glz_encode_batch(dev, items, split=bool(f % 2))
            for k, (s, n) in enumerate(zip(sessions, sizes)):
                assert n > 0
                s.check(n, "batch frame %d session %d (%s)" % (f, k, s.kind))
    finally:
        for s in sessions:
            s.close()


SMALL = [(B.LZ_IMAGE_TYPE_RGB32, 96, 40, "desktop"), (B.LZ_IMAGE_TYPE_RGB32, 64, 64, "noise"),
         (B.LZ_IMAGE_TYPE_RGBA, 80, 30, "desktop"), (B.LZ_IMAGE_TYPE_RGB16, 120, 20, "flat"),
         (B.LZ_IMAGE_TYPE_RGB24, 33, 17, "desktop"), (B.LZ_IMAGE_TYPE_RGB32, 7, 3, "noise"),
         (B.LZ_IMAGE_TYPE_RGB32, 200, 30, "flat"), (B.LZ_IMAGE_TYPE_RGB32, 131, 37, "repeat"),
         (B.LZ_IMAGE_TYPE_RGBA, 64, 33, "repeat"), (B.LZ_IMAGE_TYPE_RGB24, 50, 21, "repeat"),
         (B.LZ_IMAGE_TYPE_RGB32, 129, 1, "flat"), (B.LZ_IMAGE_TYPE_RGB32, 258, 3, "flat")]


def test_sim_batch_of_mixed_sessions(sim, oracle):
    run_batches(sim, oracle, SMALL, 4, window=1 << 15)


class Client:
    """One client with `nchan` display channels: `nchan` encoders on ONE dictionary
    (server/red_worker.c:9677-9754), every channel with its own picture sequence.
    The oracle is fed the same images in the same order through the same encoder ids."""

    def __init__(self, dev, oracle, nchan, w, h, window, seed, typ=B.LZ_IMAGE_TYPE_RGB32):
        self.dev, self.w, self.h, self.typ, self.nchan = dev, w, h, typ, nchan
        self.rng = np.random.default_rng(seed)
        self.seqs = [frames.DesktopSequence(max(w, 96), max(h, 96), seed * 10 + c) for c in range(nchan)]
        self.od = oracle.GlzDict(window, nchan)
        self.gd = B.GlzDictionary(dev, window, nchan)
        self.model = WindowModel(window, nchan)
        self.ges = [B.GlzEncoder(dev, self.gd, c) for c in range(nchan)]
        self.cap = dev.L.spice_gpu_glz_out_bound(typ, w, h)
        self.outs = [dev.malloc(self.cap) for _ in range(nchan + 1)]   # one per job of a batch
        self.held, self.px, self.tag = [], [], 0

    def items(self, order=None):
        """one job per channel, in `order` (default 0..nchan-1)"""
        self.order = list(order if order is not None else range(self.nchan))
        self.px, self.held, its = [], [], []
        for k, c in enumerate(self.order):
            px = pixels_for(self.typ, self.seqs[c].next()[:self.h, :self.w].copy(), self.rng)
            p = self.dev.to_device(px)
            self.px.append(px)
            self.held.append(p)
            its.append(dict(encoder=self.ges[c], type=self.typ, width=self.w, height=self.h,
                            stride=px.shape[1], chunks=[p], chunk_rows=[self.h], out=self.outs[k],
                            out_capacity=self.cap, usr_context=5000 + self.tag))
            self.tag += 1
        return its

    def check(self, sizes, what):
        tag0 = self.tag - len(self.order)
        for k, (c, n) in enumerate(zip(self.order, sizes)):
            want, _ = self.od.encode(self.typ, self.w, self.h, self.px[k].shape[1], [self.px[k]],
                                     encoder_id=c, usr_tag=5000 + tag0 + k)
            assert n > 0, what
            assert_same_stream(want, self.dev.download(self.outs[k], n), "%s channel %d" % (what, c))
            self.model.add(self.typ, self.h, self.px[k].shape[1], [self.h], 5000 + tag0 + k)
        assert self.od.pop_freed() == self.model.pop_freed() == self.gd.usr.freed_images, what
        self.model.check(self.gd.dump(), what)
        self.gd.usr.freed_images.clear()
        for p in self.held:
            self.dev.free(p)

    def close(self):
        for o in self.outs:
            self.dev.free(o)
        for e in self.ges:
            e.close()
        self.gd.close()
        self.od.close()


def run_shared_dictionary(dev, oracle, w, h, window, nframes, nclients=2, nchan=4):
    """batches that hold several images per dictionary: the channels of a client, in
    the caller's order, between sessions that have a dictionary to themselves"""
    clients = [Client(dev, oracle, nchan, w, h, window, seed=3 + k) for k in range(nclients)]
    solo = [Session(dev, oracle, B.LZ_IMAGE_TYPE_RGB32, w // 2 + 3, h // 2 + 1, kind, seed=90 + k,
                    window=window) for k, kind in enumerate(("desktop", "noise"))]
    try:
        for f in range(nframes):
            order = list(range(nchan))
            if f % 2:
                order = order[::-1] + [0]        # another order, and one channel twice
            groups = [c.items(order) for c in clients]
            singles = [s.next_image() for s in solo]
            # interleave: client 0's first, a solo, client 1's first, ... keeps per-dictionary order
            items, owner = [], []
            for k in range(max(len(g) for g in groups)):
                for ci, g in enumerate(groups):
                    if k < len(g):
                        items.append(g[k])
                        owner.append(("c", ci, k))
                if k < len(singles):
                    items.append(singles[k])
                    owner.append(("s", k, 0))
            sizes, _ = B.glz_encode_batch(dev, items, split=bool(f % 2))
            for ci, c in enumerate(clients):
                mine = [sizes[i] for i, o in enumerate(owner) if o[0] == "c" and o[1] == ci]
                c.check(mine, "frame %d client %d" % (f, ci))
            for k, s in enumerate(solo):
                s.check(sizes[owner.index(("s", k, 0))], "frame %d solo %d" % (f, k))
    finally:
        for c in clients:
            c.close()
        for s in solo:
            s.close()


def test_sim_shared_dictionary_batch(sim, oracle):
    # the window holds about six of the images, so every batch evicts
    run_shared_dictionary(sim, oracle, 96, 40, window=96 * 40 * 6 + 100, nframes=4)


def run_failed_batches(sim, oracle, scale=1):
    """a batch that is refused changes nothing; a job whose output does not fit leaves a
    dead image behind (as after remove_image) and the sessions carry on"""
    specs = [(B.LZ_IMAGE_TYPE_RGB32, 96 * scale, 40 * scale, "desktop"),
             (B.LZ_IMAGE_TYPE_RGB32, 64 * scale, 32 * scale, "noise"),
             (B.LZ_IMAGE_TYPE_RGB32, 80 * scale, 30 * scale, "desktop")]
    sessions = [Session(sim, oracle, *spec, seed=11 + k, window=(1 << 14) * scale * scale)
                for k, spec in enumerate(specs)]
    try:
        items = [s.next_image() for s in sessions]
        sizes, _ = B.glz_encode_batch(sim, items)
        for s, n in zip(sessions, sizes):
            s.check(n, "first batch")
        # (1) the last job is invalid: refused before any window is touched
        items = [s.next_image() for s in sessions]
        bad = dict(items[2], stride=items[2]["stride"] + 4)
        with pytest.raises(B.SpiceGpuError, match="stride"):
            B.glz_encode_batch(sim, items[:2] + [bad])
        sizes, _ = B.glz_encode_batch(sim, items)          # the same images, now accepted
        for s, n in zip(sessions, sizes):
            s.check(n, "after a refused batch")
        # (2) the noise session's output buffer is too small: that job fails on the device
        items = [s.next_image() for s in sessions]
        items[1] = dict(items[1], out_capacity=200)
        sizes, imgs = B.glz_encode_batch(sim, items, raise_on_error=False)
        assert sizes[1] < 0 and imgs[1] is None and sizes[0] > 0 and sizes[2] > 0
        sessions[0].check(sizes[0], "beside a failed job")
        sessions[2].check(sizes[2], "beside a failed job")
        s1 = sessions[1]
        _, dead = s1.od.encode(s1.typ, s1.w, s1.h, s1.stride, s1.chunks)   # the oracle: encode, then remove
        s1.od.remove_image(dead)
        for p in s1.held:
            sim.free(p)
        for f in range(3):
            items = [s.next_image() for s in sessions]
            sizes, _ = B.glz_encode_batch(sim, items)
            for s, n in zip(sessions, sizes):
                s.check(n, "batch %d after a failed job" % f)
    finally:
        for s in sessions:
            s.close()


def test_sim_failed_batches_leave_windows_usable(sim, oracle):
    run_failed_batches(sim, oracle)


@pytest.mark.gpu
def test_gpu_failed_batches_leave_windows_usable(gpu, oracle):
    run_failed_batches(gpu, oracle, scale=4)


def run_two_flights(dev, oracle, n, w, h, nframes, window):
    """Two batches in flight over disjoint dictionaries: half A's frame k+1 is submitted while
    half B's frame k still runs.  Every stream equals the oracle's, a third batch and a second
    batch on a dictionary that is already in flight are refused, and the synchronous drop-in on
    such a dictionary fails instead of racing."""
    seqs = [frames.DesktopSequence(max(w, 96), max(h, 96), seed=170 + k) for k in range(n)]
    ods = [oracle.GlzDict(window, 1) for _ in range(n)]
    gds = [B.GlzDictionary(dev, window, 1) for _ in range(n)]
    ges = [B.GlzEncoder(dev, d, 0) for d in gds]
    cap = dev.L.spice_gpu_glz_out_bound(B.LZ_IMAGE_TYPE_RGB32, w, h)
    outs = [dev.malloc(cap) for _ in range(n)]
    imgs = [[np.ascontiguousarray(s.next()[:h, :w]).reshape(h, -1) for _ in range(nframes)] for s in seqs]
    dev_imgs = [[dev.to_device(im) for im in row] for row in imgs]
    halves = [list(range(0, n // 2)), list(range(n // 2, n))]
    try:
        plans = [B.GlzBatchPlan(dev, [dict(encoder=ges[k], type=B.LZ_IMAGE_TYPE_RGB32, width=w, height=h,
                                           stride=w * 4, chunks=[dev_imgs[k][0]], chunk_rows=[h],
                                           out=outs[k], out_capacity=cap) for k in half])
                 for half in halves]
        ptrs = lambda a, f: np.array([dev_imgs[k][f] for k in halves[a]], dtype=np.uint64)

        def check(a, f, sizes):
            for j, k in enumerate(halves[a]):
                want, _ = ods[k].encode(B.LZ_IMAGE_TYPE_RGB32, w, h, w * 4, [imgs[k][f]])
                assert_same_stream(want, dev.download(outs[k], int(sizes[j])),
                                   "two flights: frame %d session %d" % (f, k))

        plans[0].submit(ptrs(0, 0))
        plans[1].submit(ptrs(1, 0))
        with pytest.raises(B.SpiceGpuError, match="in flight"):        # a third batch
            plans[0].submit(ptrs(0, 0))
        with pytest.raises(B.SpiceGpuError):                           # the synchronous drop-in on a busy dictionary
            ges[0].encode(B.LZ_IMAGE_TYPE_RGB32, w, h, w * 4, [imgs[0][0]])
        for f in range(nframes):
            for a in (0, 1):
                check(a, f, plans[a].finish())                         # the oldest batch is half a's frame f
                if f + 1 < nframes:
                    if a == 0 and f == 0:
                        # half B is still in flight: a batch that names one of its dictionaries is refused
                        mixed = B.GlzBatchPlan(dev, [dict(encoder=ges[halves[1][0]], type=B.LZ_IMAGE_TYPE_RGB32,
                                                          width=w, height=h, stride=w * 4,
                                                          chunks=[dev_imgs[halves[1][0]][1]], chunk_rows=[h],
                                                          out=outs[halves[1][0]], out_capacity=cap)])
                        with pytest.raises(B.SpiceGpuError, match="disjoint"):
                            mixed.submit()
                    plans[a].submit(ptrs(a, f + 1))
        # nothing is in flight any more: the one-call interface works as before
        with pytest.raises(B.SpiceGpuError, match="without a matching submit"):
            plans[0].finish()
    finally:
        for row in dev_imgs:
            for p in row:
                dev.free(p)
        for k in range(n):
            dev.free(outs[k])
            ges[k].close()
            gds[k].close()
            ods[k].close()


def test_sim_two_flights(sim, oracle):
    run_two_flights(sim, oracle, 4, 96, 48, 4, window=1 << 15)


@pytest.mark.gpu
def test_gpu_two_flights(gpu, oracle):
    run_two_flights(gpu, oracle, 12, 640, 400, 5, window=1 << 21)


def run_finish_thread_rule(dev, oracle):
    """a batch in flight holds its dictionaries' locks: only the submitting thread may
    finish it; another thread is refused and the batch stays in flight"""
    import threading
    sessions = [Session(dev, oracle, B.LZ_IMAGE_TYPE_RGB32, 96, 40, "desktop", seed=31 + k, window=1 << 14)
                for k in range(2)]
    try:
        for f in range(2):
            items = [s.next_image() for s in sessions]
            jobs, keep = B.glz_jobs(items)
            assert dev.L.spice_gpu_glz_submit(jobs, len(items)) == 0, dev.last_error()
            got = {}

            def other():
                got["rc"] = dev.L.spice_gpu_glz_finish(jobs, len(items))
                got["err"] = dev.last_error()

            t = threading.Thread(target=other)
            t.start()
            t.join()
            assert got["rc"] == -1 and "thread" in got["err"], got
            assert dev.L.spice_gpu_glz_finish(jobs, len(items)) == 0, dev.last_error()
            for s, j in zip(sessions, range(len(items))):
                s.check(jobs[j].out_size, "frame %d, finished by its own thread" % f)
            del keep
    finally:
        for s in sessions:
            s.close()


def test_sim_finish_thread_rule(sim, oracle):
    run_finish_thread_rule(sim, oracle)


def run_plan(dev, oracle, n, w, h, nframes, window):
    """GlzBatchPlan: the job array is built once and only the frame pointers change"""
    seqs = [frames.DesktopSequence(max(w, 96), max(h, 96), seed=70 + k) for k in range(n)]
    ods = [oracle.GlzDict(window, 1) for _ in range(n)]
    gds = [B.GlzDictionary(dev, window, 1) for _ in range(n)]
    ges = [B.GlzEncoder(dev, d, 0) for d in gds]
    cap = dev.L.spice_gpu_glz_out_bound(B.LZ_IMAGE_TYPE_RGB32, w, h)
    outs = [dev.malloc(cap) for _ in range(n)]
    imgs = [[np.ascontiguousarray(s.next()[:h, :w]).reshape(h, -1) for _ in range(nframes)] for s in seqs]
    if n > 2:                      # one session far cheaper, one far dearer than the rest
        imgs[1] = [np.zeros_like(im) for im in imgs[1]]
        rng = np.random.default_rng(5)
        imgs[2] = [rng.integers(0, 256, im.shape, dtype=np.uint8) for im in imgs[2]]
    dev_imgs = [[dev.to_device(im) for im in row] for row in imgs]
    try:
        plan = B.GlzBatchPlan(dev, [dict(encoder=ges[k], type=B.LZ_IMAGE_TYPE_RGB32, width=w, height=h,
                                         stride=w * 4, chunks=[dev_imgs[k][0]], chunk_rows=[h],
                                         out=outs[k], out_capacity=cap) for k in range(n)])
        for f in range(nframes):
            sizes = plan.run(np.array([dev_imgs[k][f] for k in range(n)], dtype=np.uint64))
            for k in range(n):
                want, _ = ods[k].encode(B.LZ_IMAGE_TYPE_RGB32, w, h, w * 4, [imgs[k][f]])
                assert_same_stream(want, dev.download(outs[k], int(sizes[k])),
                                   "plan frame %d session %d" % (f, k))
    finally:
        for row in dev_imgs:
            for p in row:
                dev.free(p)
        for k in range(n):
            dev.free(outs[k])
            ges[k].close()
            gds[k].close()
            ods[k].close()


def test_sim_batch_plan(sim, oracle):
    run_plan(sim, oracle, 4, 96, 48, 3, window=1 << 15)


@pytest.mark.gpu
def test_gpu_batch_of_mixed_sessions(gpu, oracle):
    big = SMALL + [(B.LZ_IMAGE_TYPE_RGB32, 640, 400, "desktop"), (B.LZ_IMAGE_TYPE_RGB32, 320, 200, "noise"),
                   (B.LZ_IMAGE_TYPE_RGBA, 300, 300, "desktop"), (B.LZ_IMAGE_TYPE_RGB32, 1024, 64, "flat")]
    run_batches(gpu, oracle, big, 6, window=1 << 19)


@pytest.mark.gpu
def test_gpu_finish_thread_rule(gpu, oracle):
    run_finish_thread_rule(gpu, oracle)


@pytest.mark.gpu
def test_gpu_shared_dictionary_batch(gpu, oracle):
    run_shared_dictionary(gpu, oracle, 320, 200, window=320 * 200 * 6 + 100, nframes=5)


@pytest.mark.gpu
def test_gpu_batch_plan_1080p(gpu, oracle):
    """BASELINE configs[1] through the interface bench.py measures: 24 sessions of
    1920x1080 desktop frames, 16 Mi-pixel windows"""
    run_plan(gpu, oracle, 24, 1920, 1080, 4, window=16 * 1024 * 1024)
