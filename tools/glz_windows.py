"""Window statistics of the GLZ parse from the oracle's stop trace: how many
256-position windows the device walk opens per frame and how their stops and
match lengths are distributed."""
import sys, ctypes, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from oracle import oracle as O
from spice_amd import frames
W, H = 1920, 1080
L = O.lib()
L.glzo_set_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
L.glzo_trace_count.restype = ctypes.c_size_t
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 100
nfr = int(sys.argv[2]) if len(sys.argv) > 2 else 4
seq = frames.DesktopSequence(W, H, seed=seed)
d = O.GlzDict(16 * 1024 * 1024, 1)
buf = np.zeros(2 * 4_000_000, dtype=np.uint32)
for k in range(nfr):
    f = seq.next().reshape(H, -1)
    L.glzo_set_trace(buf.ctypes.data, 4_000_000)
    d.encode(O.TYPE_RGB32, W, H, W * 4, [f])
    n = L.glzo_trace_count()
    t = buf[:2 * n].reshape(-1, 2)
    pos = t[:, 0].astype(np.int64); ln = (t[:, 1] & 0x3fffffff).astype(np.int64); rle = (t[:, 1] >> 31) & 1
    wins = []; i = 0
    while i < n:
        start = pos[i]; j = i
        while j < n and pos[j] < start + 256:
            j += 1
        last_len = ln[j - 1]; leave = pos[j - 1] + max(last_len, 1) - start
        wins.append((j - i, int(leave), int(rle[j - 1])))
        i = j
    w = np.array(wins)
    st = w[:, 0]
    print("frame", k, "stops", n, "windows", len(w), "of", W * H // 256,
          "| stops/window hist (1,2-4,5-16,17-64,65+):",
          [int(((st >= a) & (st <= b)).sum()) for a, b in ((1, 1), (2, 4), (5, 16), (17, 64), (65, 999))],
          "| single-stop windows leaving by >=256:", int(((st == 1) & (w[:, 1] >= 256)).sum()),
          "| exit jump hist (<512,<4096,<65536,more):",
          [int(((w[:, 1] >= a) & (w[:, 1] < b)).sum()) for a, b in ((0, 512), (512, 4096), (4096, 65536), (65536, 1 << 40))],
          "rle exits", int(w[:, 2].sum()))
