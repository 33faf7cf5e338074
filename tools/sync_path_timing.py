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
