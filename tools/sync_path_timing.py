"""Per-call wall time of the synchronous, host-pointer entry points (H2D of the frame,
kernels, D2H of the stream): glz_encode on 1080p frames and the MJPEG encoder on 4K frames."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import spice_amd
from spice_amd import binding as B, frames

gpu = spice_amd.load(0)
W, H = 1920, 1080
seq = frames.DesktopSequence(W, H, seed=100)
fr = [seq.next().reshape(H, -1) for _ in range(8)]
d = B.GlzDictionary(gpu, 16 << 20, 1)
e = B.GlzEncoder(gpu, d, 0)
ts = []
for f in fr:
    t = time.perf_counter()
    s, _ = e.encode(B.LZ_IMAGE_TYPE_RGB32, W, H, W * 4, [f])
    ts.append(time.perf_counter() - t)
print("glz_encode 1080p: ms per call", [round(x * 1e3, 1) for x in ts], "-> %.0f Mpx/s" % (W * H / np.median(ts[1:]) / 1e6))
# worker threads, one display each (server/red_dispatcher.c:1148), calling glz_encode side by side
import threading
for nth in (2, 4, 8, 16):
    seqs = [frames.DesktopSequence(W, H, seed=300 + t) for t in range(nth)]
    frs = [[q.next().reshape(H, -1) for _ in range(6)] for q in seqs]
    ds = [B.GlzDictionary(gpu, 16 << 20, 1) for _ in range(nth)]
    es = [B.GlzEncoder(gpu, x, 0) for x in ds]
    for t in range(nth):
        es[t].encode(B.LZ_IMAGE_TYPE_RGB32, W, H, W * 4, [frs[t][0]])
    def work(t):
        for f in frs[t][1:]:
            es[t].encode(B.LZ_IMAGE_TYPE_RGB32, W, H, W * 4, [f])
    ths = [threading.Thread(target=work, args=(t,)) for t in range(nth)]
    t0 = time.perf_counter()
    for th in ths: th.start()
    for th in ths: th.join()
    dt = time.perf_counter() - t0
    print("glz_encode 1080p, %d worker threads: %.1f ms per frame per thread, %.0f Mpx/s in all"
          % (nth, dt / 5 * 1e3, nth * 5 * W * H / dt / 1e6), flush=True)
    for x in es: x.close()
    for x in ds: x.close()
# callers of unequal size side by side: does a thread with small images pay for its
# neighbours' large ones?  (calls that arrive together share a launch: the small image's
# result comes when the launch's last image is done)
SW, SH = 320, 200
def small_frames(seed, n):
    q = frames.DesktopSequence(SW, SH, seed=seed)
    return [q.next().reshape(SH, -1) for _ in range(n)]
def timed_small(n_big, rounds=12):
    ds = [B.GlzDictionary(gpu, 16 << 20, 1) for _ in range(1 + n_big)]
    es = [B.GlzEncoder(gpu, x, 0) for x in ds]
    sf = small_frames(900, rounds + 1)
    bigs = [[q.next().reshape(H, -1) for _ in range(rounds + 1)]
            for q in (frames.DesktopSequence(W, H, seed=500 + t) for t in range(n_big))]
    es[0].encode(B.LZ_IMAGE_TYPE_RGB32, SW, SH, SW * 4, [sf[0]])
    for t in range(n_big):
        es[1 + t].encode(B.LZ_IMAGE_TYPE_RGB32, W, H, W * 4, [bigs[t][0]])
    stop = threading.Event()
    def big(t):
        k = 1
        while not stop.is_set():
            es[1 + t].encode(B.LZ_IMAGE_TYPE_RGB32, W, H, W * 4, [bigs[t][1 + (k % rounds)]])
            k += 1
    ths = [threading.Thread(target=big, args=(t,)) for t in range(n_big)]
    for th in ths: th.start()
    time.sleep(0.05 if n_big else 0.0)
    lat = []
    for f in sf[1:]:
        t0 = time.perf_counter()
        es[0].encode(B.LZ_IMAGE_TYPE_RGB32, SW, SH, SW * 4, [f])
        lat.append(time.perf_counter() - t0)
        time.sleep(0.003)
    stop.set()
    for th in ths: th.join()
    for x in es: x.close()
    for x in ds: x.close()
    return np.array(lat) * 1e3
W, H = 1920, 1080
for n_big in (0, 1, 4, 15):
    lat = timed_small(n_big)
    print("glz_encode %dx%d beside %d threads coding 1080p frames: median %.2f ms, max %.2f ms per call"
          % (SW, SH, n_big, np.median(lat), lat.max()), flush=True)
W, H = 3840, 2160
vs = frames.VideoSequence(W, H, seed=7)
vf = [vs.next().reshape(H, -1) for _ in range(5)]
m = B.MJpegEncoder(gpu)
ts = []
for f in vf:
    t = time.perf_counter()
    out = m.encode_frame(B.BITMAP_FMT_32BIT, W, H, f, W * 4)
    ts.append(time.perf_counter() - t)
print("mjpeg 4K frame: ms per call", [round(x * 1e3, 1) for x in ts], "-> %.0f Mpx/s" % (W * H / np.median(ts[1:]) / 1e6))
