#!/usr/bin/env python3
"""Throughput of the image-compression hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload NAME]

One process per GPU.  `--gpus N` with N > 1 and no WORLD_SIZE in the environment
starts the N ranks itself (a torch.distributed.run child, before this process
touches a GPU); under torch.distributed.run it is one of the ranks.  A "step" is
one pass of the hot path over one batch of synthetic input that is already
resident in HBM:

  glz1080    (BASELINE configs[1], the default)  every one of --sessions display
             sessions on this GPU encodes its next 1920x1080 RGB32 desktop frame
             with GLZ through its own 16 Mi-pixel dictionary window
  mjpeg2160  (BASELINE configs[2])  --frames 3840x2160 BGRX video frames go
             through the MJPEG encoder (IFAST quality 70 as the reference's stream
             encoder is configured; ISLOW 85 and IFAST 80 are reported beside it)
  sessions64 (BASELINE configs[3])  the GLZ half of 64 concurrent 1080p sessions,
             sharded by session over the ranks (fixed total work: strong scaling);
             the QUIC half needs spice-common's encoder, which is not built
  multimon   (BASELINE configs[4])  --clients clients with four 3840x2160 monitors
             each: per tick every monitor sends a desktop update of 70 % of its
             area through GLZ -- the four display channels of a client share ONE
             dictionary -- and a video frame of the other 30 % through MJPEG

Sessions / streams are independent, so ranks share nothing on the data path
(weak scaling: every GPU gets the same number of sessions).  RCCL is used only
for the barrier, the max-over-ranks time and the gather of the per-codec
statistics.  Rank 0 prints one JSON line.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E, datasheet
WORKLOADS = ["glz1080", "mjpeg2160", "sessions64", "multimon"]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="glz1080", choices=WORKLOADS)
    ap.add_argument("--sessions", type=int, default=2880,
                    help="GLZ display sessions per GPU (2880 x 78 MiB of window state + the input pool "
                         "fit a fresh 288 GB device; 3072 do not)")
    ap.add_argument("--flights", type=int, default=2, choices=(1, 2, 3, 4),
                    help="GLZ batches kept in flight: with 2 the sessions' two halves are submitted "
                         "alternately, each while the other still runs")
    ap.add_argument("--frames", type=int, default=64,
                    help="MJPEG frames (= streams) per step per GPU; 64 x 4K are 51 rounds of workgroups "
                         "on the GPU's 1024 slots (16 are 12.7 rounds: 10 %% more time per frame)")
    ap.add_argument("--clients", type=int, default=8, help="multimon: four-monitor clients per GPU")
    ap.add_argument("--pool", type=int, default=256, help="distinct synthetic desktop sequences per GPU")
    ap.add_argument("--out-mb", type=float, default=2.0,
                    help="GLZ output buffer per session in MiB (0 = the worst-case bound, 10 MB at 1080p)")
    ap.add_argument("--verify", type=int, default=8,
                    help="sessions whose last stream is compared with the CPU oracle after the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--also", type=int, default=1,
                    help="after glz1080 also time the other workloads (reported under 'also')")
    ap.add_argument("--sim", action="store_true",
                    help="TEST ONLY: the CPU wave-simulator build of the kernels, tiny frames, gloo "
                         "(checks the launcher and the line's shape on a machine without a GPU)")
    return ap.parse_args()


def spawn_ranks(args):
    """--gpus N without a launcher: start the N ranks as a child job and hand its exit code
    on.  Nothing in this process has touched a GPU yet."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node=%d" % args.gpus, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


class Geometry:
    """frame sizes: BASELINE's, or tiny ones for the simulator run"""

    def __init__(self, sim):
        if sim:
            self.w1080, self.h1080, self.window = 192, 96, 1 << 16
            self.w2160, self.h2160 = 160, 96
        else:
            self.w1080, self.h1080, self.window = 1920, 1080, 16 * 1024 * 1024
            self.w2160, self.h2160 = 3840, 2160


def pmc_traffic(workload, **same):
    """HBM bytes per launch of the workload's dominant kernel, from the PMC passes kept under
    profiles/ (rocprofv3 cannot run inside this process); None when they were taken on another
    configuration."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)[workload]
        return int(t["traffic_bytes"]) if all(t.get(k) == v for k, v in same.items()) else None
    except (OSError, KeyError, ValueError):
        return None


def device_free_bytes(local_rank):
    """free memory of this rank's GPU, or 0 when it cannot be told"""
    try:
        import torch
        if torch.cuda.is_available():
            return int(torch.cuda.mem_get_info(local_rank % torch.cuda.device_count())[0])
    except Exception:
        pass
    return 0


def sha(a):
    return hashlib.sha256(bytes(a)).hexdigest()


def roofline_of(kernel, alg_bytes, launches, ms, traffic=None, union_ms=None, **extra):
    """alg_bytes: algorithmic bytes of ONE launch.  With union_ms (launches of two batches in flight
    overlap on the device) the achieved rate is all launches' bytes over the time at least one of
    them was running, not bytes per launch over a launch's duration -- each of two overlapping
    launches has half the GPU, and its own duration says nothing about the kernel."""
    avg_s = ms / launches / 1e3
    overlapped = union_ms is not None and union_ms < 0.98 * ms
    ach = (alg_bytes * launches / (union_ms / 1e3) if overlapped else alg_bytes / avg_s) / 1e9
    r = {"bound": "hbm", "kernel": kernel, "achieved": round(ach, 2), "peak": HBM_PEAK_GBPS,
         "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 5), "traffic": traffic,
         "avg_launch_ms": round(ms / launches, 3), "algorithmic_bytes_per_launch": int(alg_bytes)}
    if overlapped:
        r["launches_overlap"] = {"launches": int(launches), "sum_of_durations_ms": round(ms, 3),
                                 "union_of_intervals_ms": round(union_ms, 3),
                                 "how": "achieved = launches x algorithmic_bytes_per_launch / union of the "
                                        "launches' [start, stop] intervals (HIP events on each launch's own "
                                        "stream); avg_launch_ms is the plain mean duration, as rocprofv3 shows it"}
    r.update(extra)
    return r


def glz_parse_times(times):
    """(kernel name, launches, total ms) of the GLZ parse: launches of fewer images than the GPU
    has CUs run the kernel for an under-filled GPU, the others the one for a full GPU"""
    best = ("glz_parse_kernel", 0, 0.0)
    for name in ("glz_parse_kernel", "glz_parse_lone_kernel"):
        n, ms = times.get(name, (0, 0.0))
        if ms > best[2]:
            best = (name, n, ms)
    return best


def usable_cores():
    """(cores this process may use, what limits them): the box's CPU share, not the machine's core
    count -- os.cpu_count() shows every core of the host, a GPU box gives one GPU's share of them"""
    n, why = os.cpu_count() or 1, "os.cpu_count()"
    try:
        a = len(os.sched_getaffinity(0))
        if a < n:
            n, why = a, "sched_getaffinity"
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max" and int(quota) // int(period) >= 1 and int(quota) // int(period) < n:
            n, why = int(quota) // int(period), "cgroup cpu.max"
    except (OSError, ValueError):
        pass
    share = os.environ.get("OMP_NUM_THREADS", "")
    if share.isdigit() and 0 < int(share) < n:
        n, why = int(share), "OMP_NUM_THREADS (the box's CPU share per GPU)"
    return n, why


def timed_threads(nthreads, work):
    """runs work(t) on nthreads threads; returns (wall seconds, [thread CPU seconds])"""
    cpu = [0.0] * nthreads

    def run(t):
        c0 = time.thread_time()          # CLOCK_THREAD_CPUTIME_ID, as COMPRESS_STAT's stat_now
        work(t)
        cpu[t] = time.thread_time() - c0

    ths = [threading.Thread(target=run, args=(t,)) for t in range(nthreads)]
    t0 = time.time()
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    return time.time() - t0, cpu


# ----------------------------------------------------------------------- GLZ
class GlzWorkload:
    name = "glz1080"
    dtype = "u8"
    scaling = "weak"
    codec = 0

    def __init__(self, gpu, args, dist, geo):
        import numpy as np
        from spice_amd import binding as B
        from spice_amd import frames
        self.np, self.B, self.gpu, self.geo, self.args = np, B, gpu, geo, args
        W, H = geo.w1080, geo.h1080
        self.W, self.H = W, H
        self.S = args.sessions
        self.probes = 0 if args.sim else 3          # single-session calls after the timed region
        self.nframes = args.warmup + args.steps + 1 + self.probes   # + the verification step
        # a session holds its window's pixels (68 MiB), its hash table (8 MiB) and an output
        # buffer of the worst-case size (10 MB): do not ask for more sessions than the GPU has room for
        free = device_free_bytes(dist.local_rank)
        pool = max(1, min(args.pool, self.S))
        self.cap = gpu.L.spice_gpu_glz_out_bound(B.LZ_IMAGE_TYPE_RGB32, W, H)
        if getattr(args, "out_mb", 0) and not args.sim:
            self.cap = min(self.cap, int(args.out_mb * (1 << 20)))
        if free:
            pool_bytes = pool * self.nframes * W * H * 4
            per_session = geo.window * 4 + geo.window // 4 + (8 << 20) + (1 << 20) + self.cap
            fit = int((free - pool_bytes - (6 << 30)) // per_session)
            if fit < self.S:
                print("bench: %d sessions do not fit %.0f GB of free device memory, running %d"
                      % (self.S, free / 1e9, max(1, fit)), file=sys.stderr, flush=True)
                self.S = max(1, fit)
                pool = max(1, min(pool, self.S))
        t0 = time.time()
        # every sequence has its own window layout, text and scroll, so sessions differ in
        # content and in cost; a sequence is shared by S / pool sessions
        self.verify_seqs = sorted(set(int(round(k * (pool - 1) / max(1, args.verify - 1)))
                                      for k in range(min(args.verify, pool))))
        self.host_frames = {}
        self.dev_frames = []
        for p in range(pool):
            seq = self.sequence(p, dist)
            fr = [seq.next() for _ in range(self.nframes)]
            if p in self.verify_seqs:
                self.host_frames[p] = fr
            self.dev_frames.append([gpu.to_device(f) for f in fr])
        self.pool = pool
        self.dicts = [B.GlzDictionary(gpu, geo.window, 1) for _ in range(self.S)]
        self.encs = [B.GlzEncoder(gpu, d, 0) for d in self.dicts]
        self.outs = [gpu.malloc(self.cap) for _ in range(self.S)]
        # the job arrays are built once; a step only swaps in every session's next frame.  With
        # two flights the sessions' halves are two batches over disjoint dictionaries: half A's
        # frame k+1 is submitted while half B's frame k still runs, so the workgroups of one
        # fill the slots the other's fast sessions have left
        nfl = max(1, min(getattr(args, "flights", 1), self.S))
        cut = [self.S * a // nfl for a in range(nfl + 1)]
        self.halves = [list(range(cut[a], cut[a + 1])) for a in range(nfl)]
        self.plans = [B.GlzBatchPlan(gpu, [self.job(s, 0) for s in half]) for half in self.halves]
        self.step_ptrs = [[np.array([self.dev_frames[s % pool][k] for s in half], dtype=np.uint64)
                           for half in self.halves] for k in range(self.nframes)]
        self.sizes = {}                 # frame number -> compressed sizes of all sessions
        self.pending = []               # (half, frame number) in submission order
        self.setup_s = time.time() - t0
        self.k = 0
        self.host_us = []
        self.units_per_step = self.S * W * H

    def sequence(self, p, dist):
        from spice_amd import frames
        return frames.DesktopSequence(self.W, self.H, seed=1000 * dist.rank + 100 + p)

    def job(self, s, k):
        B = self.B
        return dict(encoder=self.encs[s], type=B.LZ_IMAGE_TYPE_RGB32, width=self.W, height=self.H,
                    stride=self.W * 4, chunks=[self.dev_frames[s % self.pool][k]], chunk_rows=[self.H],
                    out=self.outs[s], out_capacity=self.cap)

    def _finish_oldest(self):
        a, k = self.pending.pop(0)
        got = self.plans[a].finish()
        self.sizes.setdefault(k, self.np.zeros(self.S, dtype=self.np.int32))[self.halves[a]] = got
        self.host_us[-1] += self.gpu.L.spice_gpu_glz_last_batch_host_ns() / 1e3
        self.last_sizes = self.sizes[k]
        return int(got.sum(dtype=self.np.int64))

    def step(self):
        """every session's next frame: each half's previous batch is completed (its sessions'
        last frames must be out before their next ones go in) and its next one submitted"""
        got = 0
        self.host_us.append(0.0)
        for a in range(len(self.plans)):
            if any(p[0] == a for p in self.pending):
                got += self._finish_oldest()          # the oldest batch in flight is this half's
            self.plans[a].submit(self.step_ptrs[self.k][a])
            self.pending.append((a, self.k))
        self.k += 1
        return got

    def flush(self):
        got = 0
        self.host_us.append(0.0)
        while self.pending:
            got += self._finish_oldest()
        self.host_us.pop()
        return got

    def config(self):
        return {"workload": "glz1080: 1920x1080 RGB32 desktop-like frames, GLZ, one 16 Mi-pixel "
                            "dictionary window per session (BASELINE configs[1])",
                "sessions_per_gpu": self.S, "sessions_requested": self.args.sessions,
                "frames_per_session": self.args.steps,
                "distinct_sequences_per_gpu": self.pool,
                "input_frames_shared_by_sessions": round(self.S / self.pool, 1),
                "batches_in_flight": len(self.plans),
                "out_buffer_bytes_per_session": self.cap,
                "parallelism": "session-sharded"}

    def roofline(self, times, steps, out_bytes_timed):
        kernel, launches, ms = glz_parse_times(times)
        if not launches:
            return None
        # dominant kernel: the parse.  Per launch it must read every pixel of
        # the new images (4 B/px) and write the compressed streams; hash-table
        # and reference traffic is not counted.
        alg = (self.S * self.W * self.H * 4 * steps + out_bytes_timed) / launches
        r = roofline_of(kernel, alg, launches, ms,
                        union_ms=getattr(self, "unions", {}).get(kernel),
                        traffic=pmc_traffic(self.name, sessions_per_gpu=self.S,
                                            distinct_sequences_per_gpu=self.pool,
                                            batches_in_flight=len(self.plans)),
                        sessions_per_launch=round(self.S * steps / launches, 1))
        cl, cms = times.get("glz_copy_kernel", (0, 0.0))
        if cl:
            calg = self.S * self.W * self.H * 8
            # the copy into the window arenas runs on a second stream beside the parse
            # (it only serves later images), so its duration is mostly hidden
            r["window_copy_kernel"] = {"achieved": round(calg / (cms / cl / 1e3) / 1e9, 1),
                                       "avg_launch_ms": round(cms / cl, 3), "unit": "GB/s",
                                       "runs_beside_parse": True}
        return r

    def step_compulsory_bytes(self, out_bytes_per_step):
        """4 B/px read + 4 B/px written into the HBM window + the streams; table traffic not counted"""
        return self.S * self.W * self.H * 8 + out_bytes_per_step

    def after_timed(self):
        """Outside the timed region: one more step whose streams are compared with the CPU
        oracle's for sessions of --verify distinct sequences (a stream depends on the whole
        history of its window, so the last one vouches for the ones before), then a few
        single-session calls for the latency of a session that has the GPU to itself."""
        from oracle import oracle as O
        np, B = self.np, self.B
        extra = {"host_us_per_step": round(float(np.mean(self.host_us[self.args.warmup:])), 1)}
        if not self.verify_seqs:          # --verify 0: nothing beyond the timed launches (profiling runs)
            return extra
        self.step()
        self.flush()
        nver = 0
        for p in self.verify_seqs:
            od = O.GlzDict(self.geo.window, 1)
            for k in range(self.k):
                want, _ = od.encode(O.TYPE_RGB32, self.W, self.H, self.W * 4,
                                    [self.host_frames[p][k].reshape(self.H, -1)])
            od.close()
            for s in range(p, self.S, self.pool)[:2]:      # two of the sessions that share the sequence
                got = self.gpu.download(self.outs[s], int(self.last_sizes[s]))
                if sha(got) != sha(want):
                    raise SystemExit("bench: session %d's stream differs from the CPU oracle's" % s)
                nver += 1
        extra["verified_sessions"] = nver
        if self.probes:
            lat = []
            for s in self.verify_seqs[:4]:
                for k in range(self.k, self.k + self.probes):
                    t0 = time.perf_counter()
                    B.glz_encode_batch(self.gpu, [self.job(s, k)])
                    lat.append((time.perf_counter() - t0) * 1e3)
            extra["lone_session_frame_ms"] = {"mean": round(float(np.mean(lat)), 2),
                                              "min": round(min(lat), 2), "max": round(max(lat), 2),
                                              "what": "one 1080p frame of one session, nothing else on the GPU"}
        return extra

    def end_to_end(self, steps=3):
        """pinned host frames -> H2D -> the same batch -> compressed streams D2H into pinned
        memory; PCIe-inclusive Mpixels/s.  Twice: step by step with nothing overlapped, and with
        two sets of sessions in flight -- while one set's batch is parsed, the other set's
        frames cross the link (its H2D copy and the D2H of its previous streams run on a copy
        stream of their own)."""
        import torch
        np = self.np
        B = self.B
        n = min(self.S, 256)                 # 256 sessions x 8.3 MB = 2.1 GB per step over PCIe
        half = n // 2
        nbytes = self.W * self.H * 4
        # every session sends the successive frames of one desktop sequence (a frame crosses the
        # link once per session: 256 copies of 8.3 MB from one pinned source)
        src_frames = self.host_frames[self.verify_seqs[0]]
        nsrc = min(len(src_frames), steps + 2)
        host = torch.empty((nsrc, nbytes), dtype=torch.uint8).pin_memory()
        for k in range(nsrc):
            host.numpy()[k] = src_frames[k].reshape(-1)
        dev = torch.empty((n, nbytes), dtype=torch.uint8, device="cuda")
        outs = torch.empty((n, self.cap), dtype=torch.uint8, device="cuda")
        out_host = torch.empty(n * (1 << 20), dtype=torch.uint8).pin_memory()

        def sessions(count):
            ds = [B.GlzDictionary(self.gpu, self.geo.window, 1) for _ in range(count)]
            return ds, [B.GlzEncoder(self.gpu, d, 0) for d in ds]

        def plan_of(encs, first):
            return B.GlzBatchPlan(self.gpu, [dict(encoder=encs[s], type=B.LZ_IMAGE_TYPE_RGB32, width=self.W,
                                                   height=self.H, stride=self.W * 4,
                                                   chunks=[dev[first + s].data_ptr()], chunk_rows=[self.H],
                                                   out=outs[first + s].data_ptr(), out_capacity=self.cap)
                                              for s in range(len(encs))])

        def release(ds, encs):
            for e in encs:
                e.close()
            for d in ds:
                d.close()

        def send(lo, hi, k):
            for s in range(lo, hi):
                dev[s].copy_(host[k % nsrc], non_blocking=True)

        def fetch(lo, hi, sizes):
            off = lo * (1 << 20)
            for s in range(lo, hi):
                m = int(sizes[s - lo])
                out_host[off:off + m].copy_(outs[s, :m], non_blocking=True)
                off += m

        # (1) serial: upload, parse, download, one after the other; the first step (every
        # session's first frame, an empty window) is not counted
        ds, encs = sessions(n)
        plan = plan_of(encs, 0)
        dts = []
        for k in range(steps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            send(0, n, k)
            torch.cuda.synchronize()
            sizes = plan.run()
            fetch(0, n, sizes)
            torch.cuda.synchronize()
            dts.append(time.perf_counter() - t0)
        release(ds, encs)
        serial = sum(dts[1:]) / len(dts[1:])

        # (2) overlapped: two halves, each with its own batch in flight and its own copies
        ds, encs = sessions(n)
        plans = [plan_of(encs[:half], 0), plan_of(encs[half:], half)]
        ranges = [(0, half), (half, n)]
        copy_stream = torch.cuda.Stream()
        up = [torch.cuda.Event(), torch.cuda.Event()]

        def upload(a, k):
            with torch.cuda.stream(copy_stream):
                send(ranges[a][0], ranges[a][1], k)
                up[a].record(copy_stream)

        rounds = 2 * (steps + 1)
        torch.cuda.synchronize()
        upload(0, 0)
        t_mark = None
        for r in range(rounds):
            a, k = r & 1, r >> 1
            if r == 2:                      # both halves' first frames are through: the clock starts
                torch.cuda.synchronize()
                t_mark = time.perf_counter()
            up[a].synchronize()             # this half's frames have arrived
            plans[a].submit()
            if r + 1 < rounds:
                upload(a ^ 1, (r + 1) >> 1) # the other half's next frames cross the link meanwhile
            sizes = plans[a].finish()
            with torch.cuda.stream(copy_stream):
                fetch(ranges[a][0], ranges[a][1], sizes)
        torch.cuda.synchronize()
        overlapped = (time.perf_counter() - t_mark) / ((rounds - 2) / 2)   # per step of all n sessions
        release(ds, encs)
        del host, dev, outs, out_host
        torch.cuda.empty_cache()
        link = 53.0     # GB/s, what this box's host-to-device copies reach
        return {"value": round(n * self.W * self.H / overlapped / 1e6, 1), "unit": "Mpixels/s",
                "ms_per_step": round(overlapped * 1e3, 2), "sessions": n,
                "serial": {"value": round(n * self.W * self.H / serial / 1e6, 1), "ms_per_step": round(serial * 1e3, 2)},
                "pcie_bound": {"link_GBps": link, "value": round(link * 1e9 / 4 / 1e6, 1),
                               "what": "what the link alone allows at 4 bytes per pixel"},
                "what": "pinned host -> H2D -> parse -> compressed streams D2H.  value: two halves of the "
                        "sessions in flight, one half's frames crossing the link while the other's batch "
                        "is parsed; serial: one batch at a time, nothing overlapped.  Every session sends "
                        "the successive frames of one desktop sequence; a step's first (empty windows) is "
                        "not counted"}

    def cpu_baseline(self):
        """The oracle (a scalar port of the reference GLZ encoder) on host threads, one thread per
        session as the reference runs one worker thread per display; every encode is timed with
        the thread's CPU clock, the COMPRESS_STAT method (server/red_worker.c:170-176,246-257)."""
        from oracle import oracle as O
        from spice_amd import frames
        W, H = self.W, self.H
        cores, why = usable_cores()
        threads = max(1, min(self.S, cores))     # the reference's model: one worker thread per session
        nfr = 64
        nseq = min(threads, 16)                  # sequences are shared beyond 16 threads (530 MB each)
        seqs = []
        for t in range(nseq):
            seq = frames.DesktopSequence(W, H, seed=5000 + t)
            seqs.append([seq.next().reshape(H, -1) for _ in range(nfr)])
        O.lib()

        def run(nthreads, passes):
            dicts = [O.GlzDict(self.geo.window, 1) for _ in range(nthreads)]
            per = [[] for _ in range(nthreads)]

            def work(t):
                for _ in range(passes):
                    for f in seqs[t % nseq]:
                        c0 = time.thread_time()
                        dicts[t].encode(O.TYPE_RGB32, W, H, W * 4, [f])
                        per[t].append(time.thread_time() - c0)

            wall, _ = timed_threads(nthreads, work)
            for d in dicts:
                d.close()
            enc = [x for p in per for x in p]
            return wall, enc

        wall1, enc1 = run(1, 3)                      # one session alone on one core
        walln, encn = run(threads, 4)                # about 20 s of CPU work in all
        px = W * H
        per_core_busy = len(encn) * px / sum(encn) / 1e6
        return {"value": round(threads * 4 * nfr * px / walln / 1e6, 1), "unit": "Mpixels/s",
                "cores": threads, "host_cores": os.cpu_count(), "usable_cores": cores,
                "usable_cores_limited_by": why, "kind": "port",
                "one_core": {"value": round(len(enc1) * px / sum(enc1) / 1e6, 1), "unit": "Mpixels/s",
                             "frame_ms": {"min": round(min(enc1) * 1e3, 2), "max": round(max(enc1) * 1e3, 2),
                                          "mean": round(sum(enc1) / len(enc1) * 1e3, 2)}},
                "per_core_when_all_busy": round(per_core_busy, 1),
                "timer": "CLOCK_THREAD_CPUTIME_ID per encode",
                "sample": "%d sessions x %d 1920x1080 desktop frames through the CPU oracle "
                          "(oracle/glz_oracle.c), one thread per session = min(sessions, usable cores), "
                          "%.1f s wall; before it one session alone, %d frames, %.1f s"
                          % (threads, 4 * nfr, walln, 3 * nfr, wall1)}

    def close(self):
        for e in self.encs:
            e.close()
        for d in self.dicts:
            d.close()
        for o in self.outs:
            self.gpu.free(o)
        for fr in self.dev_frames:
            for f in fr:
                self.gpu.free(f)
        self.encs, self.dicts, self.outs, self.dev_frames = [], [], [], []


# ---------------------------------------------------------------- sessions64
class Sessions64Workload(GlzWorkload):
    """BASELINE configs[3]: the GLZ half (32) of 64 concurrent 1080p sessions, every one with its own
    sequence, sharded over the ranks by session id; the total is fixed (strong scaling)."""
    name = "sessions64"
    scaling = "strong"

    def __init__(self, gpu, args, dist, geo):
        import copy
        from spice_amd import frames
        from spice_amd.sharding import sessions_of
        self.gen = frames.Sessions64(geo.w1080, geo.h1080)
        self.all_ids = self.gen.session_ids("glz")
        self.my_ids = sessions_of(dist.rank, dist.world, self.all_ids)
        a = copy.copy(args)
        a.sessions = max(1, len(self.my_ids))
        a.pool = a.sessions                      # every session its own sequence
        a.verify = min(args.verify, 4)
        super().__init__(gpu, a, dist, geo)
        self.args = args
        if not self.my_ids:                      # more ranks than sessions
            self.units_per_step = 0

    def sequence(self, p, dist):
        return self.gen.sequence(self.my_ids[p] if self.my_ids else self.all_ids[0])

    def config(self):
        return {"workload": "sessions64: 64 concurrent 1920x1080 display sessions sharded by session "
                            "(BASELINE configs[3]); the 32 GLZ sessions run, the 32 that the reference "
                            "codes with QUIC need spice-common's encoder, which is not built",
                "glz_sessions_total": len(self.all_ids), "glz_sessions_this_gpu": len(self.my_ids),
                "quic_sessions": "blocked: quic_encode is not in the reference tree",
                "parallelism": "session-sharded, fixed total"}


# --------------------------------------------------------------------- MJPEG
class MjpegWorkload:
    name = "mjpeg2160"
    dtype = "u8"
    scaling = "weak"
    codec = 1

    def __init__(self, gpu, args, dist, geo, dct=1, quality=70, shared=None):
        import numpy as np
        from spice_amd import binding as B
        from spice_amd import frames
        self.np, self.B, self.gpu, self.geo, self.args = np, B, gpu, geo, args
        self.dct, self.quality = dct, quality
        self.W, self.H = geo.w2160, geo.h2160
        self.F = args.frames
        # every frame of a step is a buffer of its own with content of its own: 64 x 33 MB are
        # eight times the 256 MiB Infinity Cache, so what the strip kernel reads comes from HBM
        pool = max(1, self.F)
        t0 = time.time()
        if shared is not None:                   # another quality over the same frames
            self.host_frames, self.dev_frames, self.owns = shared.host_frames, shared.dev_frames, False
        else:
            def make(f):                         # frame f of the stream seeded for it
                seq = frames.VideoSequence(self.W, self.H, seed=7 + 1000 * dist.rank + f)
                seq.k = f
                return seq.next()
            if pool > 4 and not args.sim:
                from concurrent.futures import ThreadPoolExecutor
                with ThreadPoolExecutor(max(1, min(8, usable_cores()[0]))) as ex:
                    self.host_frames = list(ex.map(make, range(pool)))
            else:
                self.host_frames = [make(f) for f in range(pool)]
            self.dev_frames = [gpu.to_device(f) for f in self.host_frames]
            self.owns = True
        self.pool = len(self.dev_frames)
        self.cap = max(4096, self.W * self.H)  # 1 byte per pixel is far above a frame of this content
        self.outs = [gpu.malloc(self.cap) for _ in range(self.F)]
        self.outs2 = [gpu.malloc(self.cap) for _ in range(self.F)]   # two batches are kept in flight
        self.setup_s = time.time() - t0
        self.k = 0
        self.plan = None
        self.pending = []
        self.units_per_step = self.F * self.W * self.H

    @property
    def kernel(self):
        return "jpeg_strip_kernel_ifast" if self.dct else "jpeg_strip_kernel_islow"

    def step(self):
        # the job array is built once; a step hands every stream its next frame
        np, B = self.np, self.B
        if self.plan is None:
            self.plan = B.JpegBatchPlan(self.gpu, [
                dict(pixels=self.dev_frames[f % self.pool], format=B.BITMAP_FMT_32BIT,
                     width=self.W, height=self.H, stride=self.W * 4, quality=self.quality, dct=self.dct,
                     out=self.outs[f], out_capacity=self.cap) for f in range(self.F)], alt_outs=self.outs2)
            self.step_pixels = [np.array([self.dev_frames[(f + k) % self.pool] for f in range(self.F)],
                                         dtype=np.uint64) for k in range(self.pool)]
        # A stream's next frame does not wait for the last one's bytes: the batch is started,
        # and the one before it is completed while this one runs (two in flight, each on its
        # own stream).  flush() completes the last; the timed region ends after it.
        got = 0
        if self.plan.in_flight() == 2:
            got = self._finish_one()
        self.plan.submit(self.step_pixels[self.k % self.pool])
        self.pending.append(self.k % self.pool)
        self.k += 1
        return got

    def _finish_one(self):
        self.last_sizes = self.plan.finish()
        self.last_k = self.pending.pop(0)
        self.last_outs = self.plan.last_outs or self.outs
        return int(self.last_sizes.sum())

    def flush(self):
        got = 0
        while self.plan is not None and self.plan.in_flight():
            got += self._finish_one()
        return got

    def config(self):
        return {"workload": "mjpeg2160: 3840x2160 BGRX video frames, MJPEG (%s DCT, quality %d, 4:2:0)%s "
                            "(BASELINE configs[2])"
                            % ("IFAST" if self.dct else "ISLOW", self.quality,
                               " as the reference's stream encoder runs libjpeg" if self.dct and self.quality == 70
                               else ""),
                "frames_per_step_per_gpu": self.F, "distinct_frames_per_gpu": self.pool,
                "parallelism": "stream-sharded"}

    def roofline(self, times, steps, out_bytes_timed):
        launches, ms = times.get(self.kernel, (0, 0.0))
        if not launches:
            return None
        # dominant kernel: pixels in (4 B/px) + Huffman bits out; everything
        # between stays in LDS / registers
        alg = self.F * self.W * self.H * 4 + out_bytes_timed / max(1, steps)
        main = self.dct and self.quality == 70
        r = roofline_of(self.kernel, alg, launches, ms,
                        traffic=pmc_traffic("mjpeg2160", frames_per_step_per_gpu=self.F,
                                            distinct_frames_per_gpu=self.pool) if main else None)
        # offsets + gather + finish, timed as one group
        tail = times.get("jpeg_bitstream_kernels", (0, 0.0))[1]
        r["bitstream_kernels_ms_per_launch"] = round(tail / launches, 3)
        return r

    def after_timed(self):
        """the last step's first frames against libjpeg itself (oracle/_ref) or its restatement"""
        from oracle import oracle as O
        use_ref = O.ref() is not None
        n = 0
        for f in range(min(2, self.F)):
            px = self.host_frames[(f + self.last_k) % self.pool].reshape(self.H, -1)
            if use_ref:
                want = O.ref_jpeg_encode(px, self.W, self.H, self.W * 4, O.FMT_32BIT, self.quality,
                                         O.DCT_IFAST if self.dct else O.DCT_ISLOW, 0)
            else:
                want = O.jpeg_encode(px, self.W, self.H, self.W * 4, O.FMT_32BIT, self.quality,
                                     O.DCT_IFAST if self.dct else O.DCT_ISLOW)
            got = self.gpu.download(self.last_outs[f], int(self.last_sizes[f]))
            if sha(got) != sha(want):
                raise SystemExit("bench: MJPEG frame %d differs from libjpeg's" % f)
            n += 1
        return {"verified_frames": n, "verified_against": "IJG libjpeg 9d" if use_ref else "oracle/jpeg_oracle.c"}

    def end_to_end(self, steps=3):
        """pinned host frames -> H2D -> kernels -> JFIF bytes D2H: serial, and with the two halves
        of the frames in flight (one half's pixels cross the link while the other half is coded)"""
        import torch
        B = self.B
        F, half = self.F, self.F // 2
        nbytes = self.W * self.H * 4
        host = torch.empty((F, nbytes), dtype=torch.uint8).pin_memory()
        for f in range(F):
            host.numpy()[f] = self.host_frames[f % self.pool].reshape(-1)
        dev = torch.empty((F, nbytes), dtype=torch.uint8, device="cuda")
        out_host = torch.empty(F * self.cap, dtype=torch.uint8).pin_memory()
        outs = torch.empty((F, self.cap), dtype=torch.uint8, device="cuda")

        def plan_of(lo, hi):
            return B.JpegBatchPlan(self.gpu, [
                dict(pixels=dev[f].data_ptr(), format=B.BITMAP_FMT_32BIT, width=self.W, height=self.H,
                     stride=self.W * 4, quality=self.quality, dct=self.dct, out=outs[f].data_ptr(),
                     out_capacity=self.cap) for f in range(lo, hi)])

        plan = plan_of(0, F)
        dts = []
        for _ in range(steps + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dev.copy_(host, non_blocking=True)
            torch.cuda.synchronize()
            sizes = plan.run()
            flat = torch.cat([outs[f, :int(sizes[f])] for f in range(F)])
            out_host[:flat.numel()].copy_(flat, non_blocking=True)
            torch.cuda.synchronize()
            dts.append(time.perf_counter() - t0)
        serial = min(dts[1:])

        plans = [plan_of(0, half), plan_of(half, F)]
        ranges = [(0, half), (half, F)]
        copy_stream = torch.cuda.Stream()
        up = [torch.cuda.Event(), torch.cuda.Event()]

        def upload(a):
            lo, hi = ranges[a]
            with torch.cuda.stream(copy_stream):
                dev[lo:hi].copy_(host[lo:hi], non_blocking=True)
                up[a].record(copy_stream)

        def download(a, sizes):
            lo, hi = ranges[a]
            with torch.cuda.stream(copy_stream):
                for f in range(lo, hi):
                    m = int(sizes[f - lo])
                    out_host[f * self.cap:f * self.cap + m].copy_(outs[f, :m], non_blocking=True)

        rounds = 2 * (steps + 1)
        torch.cuda.synchronize()
        upload(0)
        t_mark = None
        for r in range(rounds):
            a = r & 1
            if r == 2:
                t_mark = time.perf_counter()
            up[a].synchronize()
            upload(a ^ 1)                   # the copy engine works while this half is coded
            sizes = plans[a].run()
            download(a, sizes)
        torch.cuda.synchronize()
        overlapped = (time.perf_counter() - t_mark) / ((rounds - 2) / 2)
        del host, dev, outs, out_host, flat
        torch.cuda.empty_cache()
        link = 53.0
        return {"value": round(F * self.W * self.H / overlapped / 1e6, 1), "unit": "Mpixels/s",
                "ms_per_step": round(overlapped * 1e3, 2),
                "serial": {"value": round(F * self.W * self.H / serial / 1e6, 1), "ms_per_step": round(serial * 1e3, 2)},
                "pcie_bound": {"link_GBps": link, "value": round(link * 1e9 / 4 / 1e6, 1),
                               "what": "what the link alone allows at 4 bytes per pixel"},
                "what": "pinned host frames -> H2D -> kernels -> JFIF bytes D2H.  value: the two halves of "
                        "the frames in flight, one half's pixels crossing the link while the other is "
                        "coded; serial: nothing overlapped"}

    def cpu_baseline(self, frames_per_thread=8):
        """The real libjpeg driven as server/mjpeg_encoder.c drives it (oracle/_ref), else the
        oracle's restatement of it; one thread per stream, every frame timed with the thread's
        CPU clock."""
        from oracle import oracle as O
        cores, why = usable_cores()
        threads = max(1, min(self.F, cores))     # one thread per stream
        use_ref = O.ref() is not None
        W, H = self.W, self.H
        px = [f.reshape(H, -1) for f in self.host_frames]
        dct = O.DCT_IFAST if self.dct else O.DCT_ISLOW

        def encode(f):
            if use_ref:
                O.ref_jpeg_encode(f, W, H, W * 4, O.FMT_32BIT, self.quality, dct, 0)
            else:
                O.jpeg_encode(f, W, H, W * 4, O.FMT_32BIT, self.quality, dct)

        def run(nthreads, n):
            per = [[] for _ in range(nthreads)]

            def work(t):
                for k in range(n):
                    c0 = time.thread_time()
                    encode(px[(t + k) % len(px)])
                    per[t].append(time.thread_time() - c0)

            wall, _ = timed_threads(nthreads, work)
            return wall, [x for p in per for x in p]

        wall1, enc1 = run(1, 4)
        walln, encn = run(threads, frames_per_thread)
        return {"value": round(threads * frames_per_thread * W * H / walln / 1e6, 1),
                "unit": "Mpixels/s", "cores": threads, "host_cores": os.cpu_count(), "usable_cores": cores,
                "usable_cores_limited_by": why, "kind": "reference" if use_ref else "port",
                "one_core": {"value": round(len(enc1) * W * H / sum(enc1) / 1e6, 1), "unit": "Mpixels/s",
                             "frame_ms": round(sum(enc1) / len(enc1) * 1e3, 1)},
                "per_core_when_all_busy": round(len(encn) * W * H / sum(encn) / 1e6, 1),
                "timer": "CLOCK_THREAD_CPUTIME_ID per frame",
                "sample": "%d threads x %d 3840x2160 frames through %s, %.1f s wall; before it one "
                          "thread alone, 4 frames"
                          % (threads, frames_per_thread,
                             "IJG libjpeg 9d called as server/mjpeg_encoder.c:771-823 does"
                             if use_ref else "oracle/jpeg_oracle.c", walln)}

    def close(self):
        self.flush()
        for o in self.outs + self.outs2:
            self.gpu.free(o)
        self.outs, self.outs2 = [], []

    def free_frames(self):
        if self.owns:
            for f in self.dev_frames:
                self.gpu.free(f)
            self.dev_frames = []


# ------------------------------------------------------------------ multimon
class MultimonWorkload:
    """BASELINE configs[4].  A step is one tick of every client: 4 GLZ images per client through the
    client's one dictionary (four encoders, the caller's order) and 4 MJPEG frames per client."""
    name = "multimon"
    dtype = "u8"
    scaling = "weak"
    codec = 0
    TARGET_FPS = 60

    def __init__(self, gpu, args, dist, geo):
        import numpy as np
        from spice_amd import binding as B
        from spice_amd import frames
        self.np, self.B, self.gpu, self.geo, self.args = np, B, gpu, geo, args
        self.C, self.M = args.clients, 4
        W, H = geo.w2160, geo.h2160
        self.W, self.H = W, H
        self.nframes = args.warmup + args.steps + 1
        t0 = time.time()
        self.clients = [frames.MultimonClient(W, H, self.M, seed=4000 + 100 * (dist.rank * self.C + c))
                        for c in range(self.C)]
        self.gr, self.vr = self.clients[0].glz_rows, self.clients[0].video_rows
        self.host_ticks = []                     # client 0 only, for the check after the timed region
        self.dev_glz = [[[None] * self.nframes for _ in range(self.M)] for _ in range(self.C)]
        self.dev_vid = [[[None] * self.nframes for _ in range(self.M)] for _ in range(self.C)]
        # every client's desktop regions are its own; the video regions (MJPEG keeps no state
        # between frames) are client 0's four streams, shown on every client's monitors
        for k in range(self.nframes):
            for c, cl in enumerate(self.clients):
                if c == 0:
                    tick = cl.tick()
                    self.host_ticks.append(tick)
                else:
                    tick = [(d.next(), None) for d in cl.desk]
                for m, (g, v) in enumerate(tick):
                    self.dev_glz[c][m][k] = gpu.to_device(g)
                    self.dev_vid[c][m][k] = gpu.to_device(v) if c == 0 else self.dev_vid[0][m][k]
        # all display channels of a client share one dictionary (server/red_worker.c:9677-9754)
        self.dicts = [B.GlzDictionary(gpu, geo.window, self.M) for _ in range(self.C)]
        self.encs = [[B.GlzEncoder(gpu, self.dicts[c], m) for m in range(self.M)] for c in range(self.C)]
        self.gcap = gpu.L.spice_gpu_glz_out_bound(B.LZ_IMAGE_TYPE_RGB32, W, self.gr)
        self.vcap = max(4096, W * self.vr)
        self.gouts = [[gpu.malloc(self.gcap) for _ in range(self.M)] for _ in range(self.C)]
        self.vouts = [[gpu.malloc(self.vcap) for _ in range(self.M)] for _ in range(self.C)]
        order = [(c, m) for m in range(self.M) for c in range(self.C)]   # monitor m of every client, then m + 1
        self.order = order
        self.gplan = B.GlzBatchPlan(gpu, [
            dict(encoder=self.encs[c][m], type=B.LZ_IMAGE_TYPE_RGB32, width=W, height=self.gr, stride=W * 4,
                 chunks=[self.dev_glz[c][m][0]], chunk_rows=[self.gr], out=self.gouts[c][m],
                 out_capacity=self.gcap) for c, m in order])
        self.vplan = B.JpegBatchPlan(gpu, [
            dict(pixels=self.dev_vid[c][m][0], format=B.BITMAP_FMT_32BIT, width=W, height=self.vr,
                 stride=W * 4, quality=70, dct=B.DCT_IFAST, out=self.vouts[c][m], out_capacity=self.vcap)
            for c, m in order])
        self.gptrs = [np.array([self.dev_glz[c][m][k] for c, m in order], dtype=np.uint64)
                      for k in range(self.nframes)]
        self.vptrs = [np.array([self.dev_vid[c][m][k] for c, m in order], dtype=np.uint64)
                      for k in range(self.nframes)]
        self.setup_s = time.time() - t0
        self.k = 0
        self.glz_ms, self.mjpeg_ms = [], []
        self.units_per_step = self.C * self.M * W * H

    def step(self):
        t0 = time.perf_counter()
        self.gsizes = self.gplan.run(self.gptrs[self.k])
        t1 = time.perf_counter()
        self.vsizes = self.vplan.run(self.vptrs[self.k])
        t2 = time.perf_counter()
        self.glz_ms.append((t1 - t0) * 1e3)
        self.mjpeg_ms.append((t2 - t1) * 1e3)
        self.k += 1
        return int(self.gsizes.sum(dtype=self.np.int64)) + int(self.vsizes.sum(dtype=self.np.int64))

    def config(self):
        return {"workload": "multimon: clients with four 3840x2160 monitors; per tick and monitor a "
                            "3840x%d desktop update through GLZ (the client's four display channels "
                            "share ONE 16 Mi-pixel dictionary, encoded in channel order) and a 3840x%d "
                            "video frame through MJPEG IFAST q70 (BASELINE configs[4])" % (self.gr, self.vr),
                "clients_per_gpu": self.C, "monitors_per_client": self.M,
                "offered_fps_target": self.TARGET_FPS, "parallelism": "client-sharded"}

    def roofline(self, times, steps, out_bytes_timed):
        kernel, launches, ms = glz_parse_times(times)
        if not launches:
            return None
        # M launches per tick (one per display channel of every client), each over C images
        alg = self.C * self.W * self.gr * 4 + float(self.gsizes.sum()) / self.M
        return roofline_of(kernel, alg, launches, ms, launches_per_step=launches // max(1, steps))

    def after_timed(self):
        """fps per channel against the target, and client 0's last tick (GLZ streams of all four
        channels after the whole history, MJPEG frames) against the CPU oracle / libjpeg"""
        from oracle import oracle as O
        np = self.np
        w = self.args.warmup
        tick_ms = float(np.mean(self.glz_ms[w:])) + float(np.mean(self.mjpeg_ms[w:]))
        extra = {"tick_ms": round(tick_ms, 2), "glz_ms_per_tick": round(float(np.mean(self.glz_ms[w:])), 2),
                 "mjpeg_ms_per_tick": round(float(np.mean(self.mjpeg_ms[w:])), 3),
                 "achieved_fps_per_channel": round(1e3 / tick_ms, 2), "target_fps": self.TARGET_FPS,
                 "meets_target": bool(1e3 / tick_ms >= self.TARGET_FPS)}
        self.step()
        od = O.GlzDict(self.geo.window, self.M)
        for k in range(self.k):
            for m in range(self.M):
                g = self.host_ticks[k][m][0]
                want, _ = od.encode(O.TYPE_RGB32, self.W, self.gr, self.W * 4, [g.reshape(self.gr, -1)],
                                    encoder_id=m)
        # `want` is channel M-1's stream of the last tick: it has every earlier image behind it
        j = self.order.index((0, self.M - 1))
        got = self.gpu.download(self.gouts[0][self.M - 1], int(self.gsizes[j]))
        od.close()
        if sha(got) != sha(want):
            raise SystemExit("bench: multimon GLZ stream differs from the CPU oracle's")
        v = self.host_ticks[self.k - 1][0][1].reshape(self.vr, -1)
        if O.ref() is not None:
            wantv = O.ref_jpeg_encode(v, self.W, self.vr, self.W * 4, O.FMT_32BIT, 70, O.DCT_IFAST, 0)
        else:
            wantv = O.jpeg_encode(v, self.W, self.vr, self.W * 4, O.FMT_32BIT, 70, O.DCT_IFAST)
        gotv = self.gpu.download(self.vouts[0][0], int(self.vsizes[self.order.index((0, 0))]))
        if sha(gotv) != sha(wantv):
            raise SystemExit("bench: multimon MJPEG frame differs from libjpeg's")
        extra["verified"] = "client 0: last GLZ stream of channel 3 (whole history) and an MJPEG frame"
        return extra

    def close(self):
        for c in range(self.C):
            for m in range(self.M):
                self.encs[c][m].close()
                self.gpu.free(self.gouts[c][m])
                self.gpu.free(self.vouts[c][m])
                for k in range(self.nframes):
                    self.gpu.free(self.dev_glz[c][m][k])
                    if c == 0:
                        self.gpu.free(self.dev_vid[c][m][k])
            self.dicts[c].close()


def graduality_line(gpu, mj, args):
    """The codec-choice scorer (_get_bitmap_graduality_level) over the MJPEG workload's frames."""
    from spice_amd import binding as B
    n, reps = 64, 10
    W, H = mj.W, mj.H
    items = [dict(format=B.BITMAP_FMT_32BIT, width=W, chunks=[(mj.dev_frames[k % mj.pool], H)])
             for k in range(n)]
    B.graduality_batch(gpu, items)
    gpu.L.spice_gpu_timing_enable(1)
    gpu.L.spice_gpu_timing_reset()
    t0 = time.perf_counter()
    for _ in range(reps):
        res = B.graduality_batch(gpu, items)
    dt = time.perf_counter() - t0
    gpu.L.spice_gpu_timing_enable(0)
    launches, ms = gpu.kernel_times().get("gradual_kernel", (0, 0.0))
    samples = sum(r[2] for r in res) // 3
    alg = samples * 16  # four 4-byte pixels per sample; sums stay in registers
    line = {"workload": "graduality of %d 3840x2160 BGRX bitmaps per call (red_worker.c:5913-5967)" % n,
            "value": round(n * reps * W * H / dt / 1e6, 1), "unit": "Mpixels/s"}
    if launches:
        line["roofline"] = roofline_of("gradual_kernel", alg, launches, ms)
    if not args.no_cpu_baseline:
        from oracle import oracle as O
        f = mj.host_frames[0].reshape(H, -1)
        t0 = time.time()
        k = 0
        while time.time() - t0 < 2.0:
            O.graduality(O.FMT_32BIT, W, [f])
            k += 1
        line["cpu_baseline"] = {"value": round(k * W * H / (time.time() - t0) / 1e6, 1),
                                "unit": "Mpixels/s", "cores": 1, "kind": "port",
                                "sample": "%d 3840x2160 bitmaps through oracle/gradual_oracle.c, one thread" % k}
    return line


def run_workload(wl, gpu, dist, steps, warmup):
    flush = getattr(wl, "flush", lambda: 0)   # completes what a pipelined workload still has in flight
    for _ in range(warmup):
        wl.step()
    flush()
    gpu.L.spice_gpu_reset_stats()
    gpu.L.spice_gpu_timing_enable(1)
    gpu.L.spice_gpu_timing_reset()
    dist.barrier()
    t0 = time.perf_counter()
    out_bytes = 0
    for _ in range(steps):
        out_bytes += wl.step()
    out_bytes += flush()
    dist.barrier()
    dt = time.perf_counter() - t0
    gpu.L.spice_gpu_timing_enable(0)
    times = gpu.kernel_times()
    wl.unions = gpu.kernel_union_ms()
    dt_max = dist.max(dt)
    total_units = dist.sum(float(wl.units_per_step * steps))
    return dt_max, total_units, out_bytes, times


def measure(cls, gpu, dist, args, geo, steps, warmup, e2e=True, cpu=True, **kw):
    """one workload from set-up to its part of the line (rank 0 fills in the rank-0-only parts)"""
    wl = cls(gpu, args, dist, geo, **kw)
    dt, units, out_bytes, times = run_workload(wl, gpu, dist, steps, warmup)
    st = gpu.stats(wl.codec)                 # the timed steps only
    gathered = dist.gather_i64([st["count"], st["orig_bytes"], st["comp_bytes"], st["total_ns"]])
    part = {"workload": wl.config()["workload"], "value": round(units / dt / 1e6, 1), "unit": "Mpixels/s",
            "steps": steps, "warmup": warmup,
            "ms_per_step": round(dt / steps * 1e3, 3), "scaling": wl.scaling, "config": wl.config()}
    extra = wl.after_timed() if hasattr(wl, "after_timed") else {}
    wl.close()          # the end-to-end pass below brings its own device state
    if dist.rank == 0:
        part["compression"] = {"images": int(sum(g[0] for g in gathered)),
                               "orig_MB": round(sum(g[1] for g in gathered) / 1e6, 1),
                               "enc_MB": round(sum(g[2] for g in gathered) / 1e6, 1)}
        part["roofline"] = wl.roofline(times, steps, out_bytes)
        if part["roofline"] and hasattr(wl, "step_compulsory_bytes"):
            # every kernel of the step against the whole step's time: what the path must move
            # (pixels in, pixels into the window, streams out) over the step's wall time
            b = wl.step_compulsory_bytes(out_bytes / max(1, steps))
            ach = b / (dt / steps) / 1e9
            part["roofline"]["whole_step"] = {"compulsory_bytes_per_step": int(b), "achieved": round(ach, 2),
                                              "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 5)}
        part["kernel_ms"] = {k: round(v[1] / max(1, v[0]), 3) for k, v in times.items() if v[0]}
        part.update(extra)
        if e2e and not args.no_end_to_end and not args.sim and hasattr(wl, "end_to_end"):
            part["end_to_end"] = wl.end_to_end()
        if cpu and not args.no_cpu_baseline and hasattr(wl, "cpu_baseline"):
            part["cpu_baseline"] = wl.cpu_baseline()
            busy = part["cpu_baseline"].get("per_core_when_all_busy")
            if busy:
                # how many of the host's cores, all busy, code what this GPU codes
                part["cpu_baseline"]["gpu_equals_cores"] = round(part["value"] / dist.world / busy, 1)
    return wl, part, (dt, units, out_bytes, times)


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started %d ranks" % (args.gpus, world))

    from spice_amd.sharding import Dist
    geo = Geometry(args.sim)
    if args.sim:
        import ctypes
        from spice_amd.binding import SpiceGpu
        dist = Dist(backend="gloo", use_cuda=False)
        gpu = SpiceGpu(ctypes.CDLL(os.path.join(ROOT, "tests", "sim", "libspicegpu_sim.so")))
        gpu.init(0)
        args.sessions, args.frames, args.clients = min(args.sessions, 3), min(args.frames, 2), 1
        args.pool, args.verify = min(args.pool, 2), 2
    else:
        dist = Dist()
        import spice_amd
        gpu = spice_amd.load(dist.local_rank)

    classes = {"glz1080": GlzWorkload, "mjpeg2160": MjpegWorkload, "sessions64": Sessions64Workload,
               "multimon": MultimonWorkload}
    wl, part, (dt, units, out_bytes, times) = measure(classes[args.workload], gpu, dist, args, geo,
                                                      args.steps, args.warmup)

    result = {
        "metric": "compressed Mpixels/s per GPU & 8-GPU node (QUIC/GLZ/MJPEG); % HBM roofline",
        "value": part["value"],
        "unit": "Mpixels/s",
        "n_gpus": dist.world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": part["ms_per_step"],
        "higher_is_better": True,
        "scaling": wl.scaling,
        "vs_baseline": None,
        "dtype": wl.dtype,
        "data": "synthetic" + (" (CPU wave simulator, test only)" if args.sim else ""),
        "config": part["config"],
    }
    if dist.rank == 0:
        for k, v in part.items():
            if k not in result and k not in ("workload",):
                result[k] = v

    if args.also and args.workload == "glz1080":
        also = {}
        half = max(2, args.steps // 2)
        # a step of the MJPEG workloads lasts under a millisecond and two batches are kept in
        # flight: over three steps the figure would be the pipeline's filling and draining
        msteps = half if args.sim else max(half, 50)
        mj, p, _ = measure(MjpegWorkload, gpu, dist, args, geo, msteps, 3)
        also["mjpeg2160"] = p
        for key, dct, q in (("mjpeg2160_islow85", 0, 85), ("mjpeg2160_ifast80", 1, 80)):
            w2, p2, _ = measure(MjpegWorkload, gpu, dist, args, geo, msteps, 3, e2e=False, dct=dct, quality=q,
                                shared=mj)
            also[key] = p2
        if dist.rank == 0:
            also["graduality2160"] = graduality_line(gpu, mj, args)
        mj.free_frames()
        for key, cls in (("sessions64", Sessions64Workload), ("multimon", MultimonWorkload)):
            w3, p3, _ = measure(cls, gpu, dist, args, geo, half, 1, e2e=False, cpu=False)
            also[key] = p3
        if dist.rank == 0:
            for p in also.values():
                p.pop("config", None)
            result["also"] = also

    if dist.rank == 0:
        print(json.dumps(result), flush=True)
    dist.close()


if __name__ == "__main__":
    main()
