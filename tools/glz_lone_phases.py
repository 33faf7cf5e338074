"""Where the steps of the kernel for an under-filled GPU (glz_lone.hip) spend their time:
S sessions x F 1080p frames through the batch interface of a library whose glz_lone.hip was
built with -DGLZ_STAMPS (tools/diag/libspicegpu_stamps.so, see tools/README.md).
    python tools/glz_lone_phases.py S F
Per frame: steps (a step walks one window while the next is scanned and the previous one written
out), how many next windows were scanned ahead, how many of those the walk then entered, steps
in which the walk had to wait for a scan, and shader-clock cycles per step of each role."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from spice_amd import binding as B, frames
S = int(sys.argv[1]) if len(sys.argv) > 1 else 1
F = int(sys.argv[2]) if len(sys.argv) > 2 else 5
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "diag", os.environ.get("GLZ_STAMPS_LIB", "libspicegpu_stamps.so")))
gpu = B.SpiceGpu(lib); gpu.init(0)
W, H = 1920, 1080
dev = []
for p in range(S):
    seq = frames.DesktopSequence(W, H, seed=int(os.environ.get("SEED0", "100")) + p)
    dev.append([gpu.to_device(seq.next()) for _ in range(F)])
dicts = [B.GlzDictionary(gpu, 16 << 20, 1) for _ in range(S)]
encs = [B.GlzEncoder(gpu, d, 0) for d in dicts]
cap = 2 << 20
outs = [gpu.malloc(cap) for _ in range(S)]
plan = B.GlzBatchPlan(gpu, [dict(encoder=encs[s], type=8, width=W, height=H, stride=W * 4, chunks=[dev[s][0]],
                                  chunk_rows=[H], out=outs[s], out_capacity=cap) for s in range(S)])
st = (ctypes.c_ulonglong * 32)()
for k in range(F):
    lib.spice_gpu_debug_lone_stamps(st, 1)
    t0 = time.perf_counter()
    sizes = plan.run(np.array([dev[s][k] for s in range(S)], dtype=np.uint64))
    dt = time.perf_counter() - t0
    lib.spice_gpu_debug_lone_stamps(st, 1)
    a = list(st)
    steps = max(1, a[5])
    print("frame %d: %.2f ms | per session: %.0f steps, scanned ahead %.0f, entered %.0f (%.0f %%), steps waiting "
          "for a scan %.0f | cycles per step: walk %.0f, scan %.0f, emit %.0f; side-by-side part %.0f, "
          "stores + fix-up part %.0f | out %.0f KB"
          % (k, dt * 1e3, steps / S, a[8] / S, a[6] / S, 100.0 * a[6] / max(1, a[8]), a[7] / S,
             a[0] / steps, a[1] / steps, a[2] / steps, a[3] / steps, a[4] / steps, sizes.mean() / 1e3), flush=True)
    print("         of the windows scanned ahead %.0f were placed where a long match was expected to end, %.0f rightly"
          % (a[20] / S, a[21] / S), flush=True)
    walked = max(1, a[5] - a[7])
    print("         inside the walk, per walked window: %.1f quarters, quarter set-up %.0f, quarter end %.0f, whole walk function %.0f"
          % (a[25] / walked, a[22] / walked, a[23] / walked, a[24] / walked), flush=True)          # steps with a walk = steps - waiting steps
    kinds = ["run leaving the window", "both nearest skipped", "table compare goes on", "window compare goes on", "table compare measured ahead"]
    print("         per walked window (%.0f): walk %.0f cycles = fast loop %.0f + general path %s + rest"
          % (walked / S, a[0] / walked, a[9] / walked,
             ", ".join("%s %.2f x %.0f" % (kinds[k], a[15 + k] / walked, a[10 + k] / max(1, a[15 + k])) for k in range(5))),
          flush=True)
