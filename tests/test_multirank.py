"""The N>1 path on two gloo ranks (CPU): session sharding, barrier, max-over-ranks
time, sum of units and the stats gather that bench.py uses with RCCL on GPUs.
Each rank encodes its own sessions with the CPU wave-simulator build of the GLZ
kernel; rank 0 checks that the union equals what one process produces."""
import json
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import ctypes, json, os, sys, time
sys.path.insert(0, %(root)r)
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
from spice_amd import binding as B
from spice_amd.sharding import Dist, sessions_of
from helpers import small_desktop

dist = Dist(backend="gloo", use_cuda=False)
gpu = B.SpiceGpu(ctypes.CDLL(os.path.join(%(root)r, "tests", "sim", "libspicegpu_sim.so")))
gpu.init(0)
clients = list(range(10))
mine = sessions_of(dist.rank, dist.world, clients)
sizes = {}
dist.barrier()
t0 = time.perf_counter()
units = 0
for c in mine:
    d = B.GlzDictionary(gpu, 1 << 16, 1)
    e = B.GlzEncoder(gpu, d, 0)
    tot = 0
    for k in range(2):
        img = small_desktop(96, 40, c * 7 + k).reshape(40, -1)
        s, _ = e.encode(B.LZ_IMAGE_TYPE_RGB32, 96, 40, 96 * 4, [img])
        tot += len(s)
        units += 96 * 40
    sizes[c] = tot
    e.close(); d.close()
dist.barrier()
dt = time.perf_counter() - t0
dt_max = dist.max(dt)
total_units = dist.sum(float(units))
st = gpu.stats(0)
gathered = dist.gather_i64([st["count"], st["orig_bytes"], st["comp_bytes"], len(mine)])
vec = [sizes.get(c, 0) for c in clients]
all_sizes = dist.gather_i64(vec)
if dist.rank == 0:
    print("RESULT " + json.dumps({"dt_max": dt_max, "dt": dt, "units": total_units,
                                  "gathered": gathered, "sizes": all_sizes, "world": dist.world}))
dist.close()
'''


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_world(world):
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER % {"root": ROOT}], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    line = [l for l in outs[0][0].splitlines() if l.startswith("RESULT ")][0]
    return json.loads(line[7:])


def test_two_ranks_match_one_process(sim):
    one = run_world(1)
    two = run_world(2)
    assert two["world"] == 2 and len(two["gathered"]) == 2
    # every client encoded by exactly one rank, with the bytes one process produces
    merged = np.array(two["sizes"]).sum(axis=0)
    counts = (np.array(two["sizes"]) > 0).sum(axis=0)
    assert (counts == 1).all()
    assert merged.tolist() == one["sizes"][0]
    # whole-job aggregates
    assert two["units"] == one["units"] == 10 * 2 * 96 * 40
    assert sum(g[0] for g in two["gathered"]) == 20          # images
    assert sum(g[3] for g in two["gathered"]) == 10          # sessions
    assert sum(g[2] for g in two["gathered"]) == int(merged.sum())
    assert two["dt_max"] >= two["dt"] > 0


def test_owner_is_stable_and_spreads():
    from spice_amd.sharding import owner_of, sessions_of
    for world in (1, 2, 4, 8):
        owners = [owner_of(c, world) for c in range(64)]
        assert all(0 <= o < world for o in owners)
        assert owners == [owner_of(c, world) for c in range(64)]
        per = [len(sessions_of(r, world, range(64))) for r in range(world)]
        assert sum(per) == 64 and min(per) >= 64 // world // 2
