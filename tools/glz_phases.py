"""Where a GLZ window's time goes, under load: S sessions x F 1080p frames through the batch
interface of a library built with -DGLZ_STAMPS (tools/diag/libspicegpu_stamps.so, see
tools/README.md), whose parse kernel adds up shader-clock cycles per phase on wavefront 0.
    python tools/glz_phases.py S F [POOL]
"""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from spice_amd import binding as B, frames
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
F = int(sys.argv[2]) if len(sys.argv) > 2 else 5
POOL = int(sys.argv[3]) if len(sys.argv) > 3 else 64
# GLZ_STAMPS_LIB names another stamped build under tools/diag/ (two builds compared on one box)
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "diag", os.environ.get("GLZ_STAMPS_LIB", "libspicegpu_stamps.so")))
gpu = B.SpiceGpu(lib); gpu.init(0)
W, H = 1920, 1080
pool = min(POOL, S)
dev = []
for p in range(pool):
    seq = frames.DesktopSequence(W, H, seed=100 + p)
    dev.append([gpu.to_device(seq.next()) for _ in range(F)])
dicts = [B.GlzDictionary(gpu, 16 << 20, 1) for _ in range(S)]
encs = [B.GlzEncoder(gpu, d, 0) for d in dicts]
cap = 2 << 20
outs = [gpu.malloc(cap) for _ in range(S)]
plan = B.GlzBatchPlan(gpu, [dict(encoder=encs[s], type=8, width=W, height=H, stride=W * 4, chunks=[dev[s % pool][0]],
                                  chunk_rows=[H], out=outs[s], out_capacity=cap) for s in range(S)])
st = (ctypes.c_ulonglong * 24)()
names = ["tile", "scan", "resolve", "emit"]
for k in range(F):
    lib.spice_gpu_debug_stamps(st, 1)
    t0 = time.perf_counter()
    sizes = plan.run(np.array([dev[s % pool][k] for s in range(S)], dtype=np.uint64))
    dt = time.perf_counter() - t0
    lib.spice_gpu_debug_stamps(st, 1)
    a = list(st)
    win = max(1, a[4])
    tot = sum(a[:4])
    print("frame %d: %.1f ms, %d windows (%.0f per session), cycles per window: %s | total %.0f | in resolve: fast loop %.0f, general %.0f (%.2f calls per window, %.0f cycles per call) | out %.0f KB per session"
          % (k, dt * 1e3, win, win / S, ", ".join("%s %.0f" % (n, a[i] / win) for i, n in enumerate(names)), tot / win,
             a[5] / win, a[6] / win, a[7] / win, a[6] / max(1, a[7]), sizes.mean() / 1e3), flush=True)
    print("         inside scan: hash+bitmap+probe issue+barrier %.0f, nearest same-hash search %.0f, table entry check + pixel requests %.0f, window compares %.0f, table compare (waits for the pixels) %.0f, run length + results %.0f"
          % tuple(a[8 + i] / win for i in range(6)), flush=True)
    print("         inside emit: table stores %.0f, run bookkeeping %.0f, token sizes %.0f, prefix sum + barrier %.0f, writes %.0f"
          % tuple(a[14 + i] / win for i in range(5)), flush=True)
