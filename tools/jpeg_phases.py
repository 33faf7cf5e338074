"""Shader-clock cycles per phase of the JPEG strip kernel (wavefront 0 of every workgroup), from a
library built with -DJPEG_STAMPS (tools/diag/libspicegpu_jstamps.so, see tools/README.md).
    python tools/jpeg_phases.py [ifast|islow] [quality] [video|desktop]"""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from spice_amd import binding as B, frames
dct = 0 if (len(sys.argv) > 1 and sys.argv[1] == "islow") else 1
q = int(sys.argv[2]) if len(sys.argv) > 2 else 70
kind = sys.argv[3] if len(sys.argv) > 3 else "video"
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "diag", "libspicegpu_jstamps.so"))
gpu = B.SpiceGpu(lib); gpu.init(0)
W, H, F = 3840, 2160, 16
seq = frames.VideoSequence(W, H, 7) if kind == "video" else frames.DesktopSequence(W, H, 7)
dev = [gpu.to_device(seq.next()) for _ in range(8)]
outs = [gpu.malloc(W * H) for _ in range(F)]
items = [dict(pixels=dev[f % 8], format=8, width=W, height=H, stride=W * 4, quality=q, dct=dct, out=outs[f], out_capacity=W * H) for f in range(F)]
st = (ctypes.c_ulonglong * 16)()
names = ["pixels->tiles", "tiles->regs", "dct", "quant", "zigzag+dc", "huff build", "huff place+flush"]
for k in range(4):
    lib.spice_gpu_debug_jpeg_stamps(st, 1)
    t0 = time.perf_counter(); sizes = B.jpeg_encode_batch(gpu, items); dt = time.perf_counter() - t0
    lib.spice_gpu_debug_jpeg_stamps(st, 1)
    a = list(st); n = max(1, a[7])
    print("%s q%d %s: %.3f ms, %d workgroups, cycles per workgroup: %s | total %.0f"
          % ("ifast" if dct else "islow", q, kind, dt * 1e3, n, ", ".join("%s %.0f" % (nm, a[i] / n) for i, nm in enumerate(names)), sum(a[:7]) / n), flush=True)
