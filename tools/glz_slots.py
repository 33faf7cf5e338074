"""How well a batch fills the GPU's workgroup slots: S sessions x F 1080p frames through the batch
interface of the diagnostic library (-DGLZ_STAMPS, tools/diag/libspicegpu_stamps.so), whose parse
kernel notes when every workgroup started and ended.  Prints, per frame: the launch's span, the
work (sum of workgroup durations) over slots x span, when the slots stop being full, and how well
the launch order (most expensive first, as the host predicts) matches the measured durations.
    python tools/glz_slots.py S F [POOL]
"""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from spice_amd import binding as B, frames
S = int(sys.argv[1]) if len(sys.argv) > 1 else 3072
F = int(sys.argv[2]) if len(sys.argv) > 2 else 4
POOL = int(sys.argv[3]) if len(sys.argv) > 3 else 256
SLOTS = 2048
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "diag", "libspicegpu_stamps.so"))
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
n = min(S, 8192)
buf = (ctypes.c_ulonglong * (2 * n))()
for k in range(F):
    t0 = time.perf_counter()
    plan.run(np.array([dev[s % pool][k] for s in range(S)], dtype=np.uint64))
    dt = time.perf_counter() - t0
    lib.spice_gpu_debug_wg_times(buf, n)
    t = np.array(list(buf), dtype=np.float64).reshape(n, 2) / 100.0   # us at 100 MHz
    start, end = t[:, 0] - t[:, 0].min(), t[:, 1] - t[:, 0].min()
    dur = end - start
    span = end.max()
    ev = np.concatenate([np.stack([start, np.ones(n)], 1), np.stack([end, -np.ones(n)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    active = np.cumsum(ev[:, 1])
    full_until = ev[np.where(active >= min(SLOTS, n) - 1)[0][-1], 0] if (active >= min(SLOTS, n) - 1).any() else 0.0
    tt = ev[:, 0]
    prof = [int(active[np.searchsorted(tt, span * f, side="right") - 1]) for f in (0.02, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9)]
    print("         workgroups running: max %d; at 2%%,10%%..90%% of the span: %s; last start at %.1f ms" % (active.max(), prof, start.max() / 1e3))
    first_round = dur[:SLOTS].mean()
    rank_corr = np.corrcoef(np.argsort(np.argsort(-dur)), np.arange(n))[0, 1]
    # what longest-first would give with the measured durations, on SLOTS machines
    import heapq
    h = [0.0] * min(SLOTS, n)
    for d in sorted(dur, reverse=True):
        heapq.heappush(h, heapq.heappop(h) + d)
    print("frame %d: %.1f ms wall; launch span %.1f ms; work/(slots x span) %.2f; slots full until %.1f ms; "
          "workgroup ms mean %.1f min %.1f max %.1f (first %d started: mean %.1f); launch order vs measured cost: rank correlation %.2f; "
          "longest-first on the measured durations: %.1f ms; work/slots %.1f ms"
          % (k, dt * 1e3, span / 1e3, dur.sum() / (min(SLOTS, n) * span), full_until / 1e3, dur.mean() / 1e3, dur.min() / 1e3,
             dur.max() / 1e3, SLOTS, first_round / 1e3, rank_corr, max(h) / 1e3, dur.sum() / min(SLOTS, n) / 1e3), flush=True)
