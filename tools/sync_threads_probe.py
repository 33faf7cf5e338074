"""Why do worker threads calling glz_encode side by side not overlap?  Same call pattern with the
frames in pageable and in pinned host memory, and the time of the C call alone."""
import os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import spice_amd
from spice_amd import binding as B, frames
gpu = spice_amd.load(0)
W, H = 1920, 1080
def run(nth, pinned):
    seqs = [frames.DesktopSequence(W, H, seed=300 + t) for t in range(nth)]
    frs = [[q.next().reshape(H, -1) for _ in range(6)] for q in seqs]
    keep = []
    if pinned:
        for t in range(nth):
            for k in range(6):
                p = torch.empty(H * W * 4, dtype=torch.uint8).pin_memory()
                p.numpy()[:] = frs[t][k].reshape(-1)
                keep.append(p)
                frs[t][k] = p.numpy().reshape(H, -1)
    ds = [B.GlzDictionary(gpu, 16 << 20, 1) for _ in range(nth)]
    es = [B.GlzEncoder(gpu, x, 0) for x in ds]
    for t in range(nth):
        es[t].encode(B.LZ_IMAGE_TYPE_RGB32, W, H, W * 4, [frs[t][0]])
    per = [[] for _ in range(nth)]
    def work(t):
        for f in frs[t][1:]:
            t0 = time.perf_counter()
            es[t].encode(B.LZ_IMAGE_TYPE_RGB32, W, H, W * 4, [f])
            per[t].append(time.perf_counter() - t0)
    ths = [threading.Thread(target=work, args=(t,)) for t in range(nth)]
    t0 = time.perf_counter()
    for th in ths: th.start()
    for th in ths: th.join()
    dt = time.perf_counter() - t0
    print("%d threads, %s frames: wall %.1f ms for 5 frames each; per call ms: %s" % (
        nth, "pinned" if pinned else "pageable", dt * 1e3,
        [round(float(np.mean(p)) * 1e3, 1) for p in per]), flush=True)
    for x in es: x.close()
    for x in ds: x.close()
for nth in (1, 2, 4):
    run(nth, False)
    run(nth, True)
