"""Per-session GLZ frame times when each session runs alone on the GPU: how uneven
the sessions of one lock-step batch are (the batch takes as long as its slowest)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spice_amd
from spice_amd import binding as B, frames
gpu = spice_amd.load(0)
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 8
NF = int(sys.argv[2]) if len(sys.argv) > 2 else 5
W, H = 1920, 1080
cap = gpu.L.spice_gpu_glz_out_bound(8, W, H)
out = gpu.malloc(cap)
rows = []
for s in range(NS):
    seq = frames.DesktopSequence(W, H, seed=100 + s)
    d = B.GlzDictionary(gpu, 16 * 1024 * 1024, 1)
    e = B.GlzEncoder(gpu, d, 0)
    ts = []
    zs = []
    for k in range(NF):
        f = gpu.to_device(seq.next())
        item = dict(encoder=e, type=8, width=W, height=H, stride=W * 4, chunks=[f], chunk_rows=[H],
                    out=out, out_capacity=cap)
        gpu.L.spice_gpu_timing_enable(1); gpu.L.spice_gpu_timing_reset()
        sizes, _ = B.glz_encode_batch(gpu, [item])
        zs.append(sizes[0] // 1000)
        ts.append(gpu.kernel_times().get("glz_parse_kernel", (0, 0.0))[1])
    rows.append(ts)
    print("seed", 100 + s, " ".join("%6.2f" % t for t in ts), "| kB", zs, flush=True)
a = np.array(rows)
print("per frame: mean", np.round(a.mean(0), 2), "max", np.round(a.max(0), 2), "mean/max", np.round(a.mean(0) / a.max(0), 2))
