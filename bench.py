#!/usr/bin/env python3
"""Throughput of the image-compression hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload glz1080|mjpeg2160]

One process per GPU (the driver launches N ranks through torch.distributed.run).
A "step" is one pass of the hot path over one batch of synthetic input that is
already resident in HBM:

  glz1080    (BASELINE configs[1], the default)  every one of --sessions display
             sessions on this GPU encodes its next 1920x1080 RGB32 desktop frame
             with GLZ through its own 16 Mi-pixel dictionary window
  mjpeg2160  (BASELINE configs[2])  --frames 3840x2160 BGRX video frames go
             through the MJPEG encoder (IFAST, quality 70, as the reference's
             stream encoder is configured)

Sessions / streams are independent, so ranks share nothing on the data path
(weak scaling: every GPU gets the same number of sessions).  RCCL is used only
for the barrier, the max-over-ranks time and the gather of the per-codec
statistics.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E, datasheet


def pmc_traffic(workload, key, value):
    """HBM bytes per launch of the workload's dominant kernel, from the PMC passes kept under
    profiles/ (rocprofv3 cannot run inside this process); None when they were taken on another
    configuration."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)[workload]
        return int(t["traffic_bytes"]) if t.get(key) == value else None
    except (OSError, KeyError, ValueError):
        return None


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="glz1080", choices=["glz1080", "mjpeg2160"])
    ap.add_argument("--sessions", type=int, default=2048, help="GLZ display sessions per GPU")
    ap.add_argument("--frames", type=int, default=16, help="MJPEG frames per step per GPU")
    ap.add_argument("--pool", type=int, default=8, help="distinct synthetic sequences per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--also-mjpeg", type=int, default=1,
                    help="after the headline workload also time mjpeg2160 (reported under 'also')")
    return ap.parse_args()


from spice_amd.sharding import Dist  # noqa: E402


# ----------------------------------------------------------------------- GLZ
W1080, H1080 = 1920, 1080
GLZ_WINDOW = 16 * 1024 * 1024


def device_free_bytes(local_rank):
    """free memory of this rank's GPU, or 0 when it cannot be told"""
    try:
        import torch
        if torch.cuda.is_available():
            return int(torch.cuda.mem_get_info(local_rank % torch.cuda.device_count())[0])
    except Exception:
        pass
    return 0


class GlzWorkload:
    name = "glz1080"
    dtype = "u8"

    def __init__(self, gpu, args, rank):
        from spice_amd import binding as B
        from spice_amd import frames
        self.B, self.gpu = B, gpu
        self.S = args.sessions
        # a session holds its window's pixels (68 MiB), its hash table (8 MiB) and an output
        # buffer of the worst-case size (10 MB): do not ask for more sessions than the GPU has room for
        free = device_free_bytes(rank)
        if free:
            pool_bytes = max(1, min(args.pool, self.S)) * (args.warmup + args.steps) * W1080 * H1080 * 4
            fit = int((free - pool_bytes - (4 << 30)) // (96 << 20))
            if fit < self.S:
                print("bench: %d sessions do not fit %.0f GB of free device memory, running %d"
                      % (self.S, free / 1e9, max(1, fit)), file=sys.stderr, flush=True)
                self.S = max(1, fit)
        self.nframes = args.warmup + args.steps
        pool = max(1, min(args.pool, self.S))
        t0 = time.time()
        self.host_frames = []
        self.dev_frames = []
        for p in range(pool):
            seq = frames.DesktopSequence(W1080, H1080, seed=1000 * rank + 100 + p)
            fr = [seq.next() for _ in range(self.nframes)]
            self.host_frames.append(fr)
            self.dev_frames.append([gpu.to_device(f) for f in fr])
        self.pool = pool
        self.dicts = [B.GlzDictionary(gpu, GLZ_WINDOW, 1) for _ in range(self.S)]
        self.encs = [B.GlzEncoder(gpu, d, 0) for d in self.dicts]
        self.cap = gpu.L.spice_gpu_glz_out_bound(B.LZ_IMAGE_TYPE_RGB32, W1080, H1080)
        self.outs = [gpu.malloc(self.cap) for _ in range(self.S)]
        # the job array is built once; a step only swaps in every session's next frame
        self.plan = B.GlzBatchPlan(gpu, [
            dict(encoder=self.encs[s], type=B.LZ_IMAGE_TYPE_RGB32, width=W1080, height=H1080,
                 stride=W1080 * 4, chunks=[self.dev_frames[s % pool][0]], chunk_rows=[H1080],
                 out=self.outs[s], out_capacity=self.cap) for s in range(self.S)])
        self.step_ptrs = [np.array([self.dev_frames[s % pool][k] for s in range(self.S)], dtype=np.uint64)
                          for k in range(self.nframes)]
        self.setup_s = time.time() - t0
        self.k = 0
        self.out_bytes = 0
        self.units_per_step = self.S * W1080 * H1080

    def step(self):
        sizes = self.plan.run(self.step_ptrs[self.k])
        self.k += 1
        self.last_out = int(sizes.sum(dtype=np.int64))
        return self.last_out

    def config(self, args):
        return {"workload": "glz1080: 1920x1080 RGB32 desktop-like frames, GLZ, one 16 Mi-pixel "
                            "dictionary window per session (BASELINE configs[1])",
                "sessions_per_gpu": self.S, "sessions_requested": args.sessions,
                "frames_per_session": args.steps,
                "distinct_sequences_per_gpu": self.pool, "parallelism": "session-sharded"}

    def roofline(self, times, steps, out_bytes_timed):
        launches, ms = times.get("glz_parse_kernel", (0, 0.0))
        if not launches:
            return None
        # dominant kernel: the parse.  Per launch it must read every pixel of
        # the new images from the window (4 B/px) and write the compressed
        # streams; hash-table and reference traffic is not counted.
        alg = self.S * W1080 * H1080 * 4 + out_bytes_timed / max(1, steps)
        avg_s = ms / launches / 1e3
        ach = alg / avg_s / 1e9
        r = {"bound": "hbm", "kernel": "glz_parse_kernel", "achieved": round(ach, 2),
             "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 5),
             "traffic": pmc_traffic("glz1080", "sessions_per_gpu", self.S),
             "avg_launch_ms": round(ms / launches, 3), "algorithmic_bytes_per_launch": int(alg)}
        cl, cms = times.get("glz_copy_kernel", (0, 0.0))
        if cl:
            calg = self.S * W1080 * H1080 * 8
            # the copy into the window arenas runs on a second stream beside the parse
            # (it only serves later images), so its duration is mostly hidden
            r["window_copy_kernel"] = {"achieved": round(calg / (cms / cl / 1e3) / 1e9, 1),
                                       "avg_launch_ms": round(cms / cl, 3), "unit": "GB/s",
                                       "runs_beside_parse": True}
        return r

    def cpu_baseline(self, args):
        """The oracle (a scalar port of the reference GLZ encoder) on host threads, one
        thread per session as the reference runs one worker thread per display."""
        from oracle import oracle as O
        from spice_amd import frames
        threads = max(1, min(16, os.cpu_count() or 1))
        nfr, passes = 64, 3     # about 25 s of CPU work in all
        seqs = []
        for t in range(threads):
            seq = frames.DesktopSequence(W1080, H1080, seed=5000 + t)
            seqs.append([seq.next().reshape(H1080, -1) for _ in range(nfr)])
        O.lib()
        dicts = [O.GlzDict(GLZ_WINDOW, 1) for _ in range(threads)]

        own = [0.0] * threads

        def work(t):
            t1 = time.time()
            for _ in range(passes):
                for f in seqs[t]:
                    dicts[t].encode(O.TYPE_RGB32, W1080, H1080, W1080 * 4, [f])
            own[t] = time.time() - t1

        ths = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
        t0 = time.time()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        dt = time.time() - t0
        one = passes * nfr * W1080 * H1080 / (sum(own) / threads) / 1e6
        return {"value": round(threads * passes * nfr * W1080 * H1080 / dt / 1e6, 1), "unit": "Mpixels/s",
                "cores": threads, "kind": "port", "per_core": round(one, 1),
                "sample": "%d sessions x %d 1920x1080 desktop frames (a 64-frame sequence, %d times) "
                          "through the CPU oracle (oracle/glz_oracle.c), one thread per session, "
                          "%.1f s wall" % (threads, passes * nfr, passes, dt)}


# --------------------------------------------------------------------- MJPEG
W2160, H2160 = 3840, 2160


class MjpegWorkload:
    name = "mjpeg2160"
    dtype = "u8"

    def __init__(self, gpu, args, rank):
        from spice_amd import binding as B
        from spice_amd import frames
        self.B, self.gpu = B, gpu
        self.F = args.frames
        pool = max(1, min(args.pool, self.F))
        t0 = time.time()
        seq = frames.VideoSequence(W2160, H2160, seed=7 + rank)
        self.host_frames = [seq.next() for _ in range(pool)]
        self.dev_frames = [gpu.to_device(f) for f in self.host_frames]
        self.pool = pool
        self.cap = W2160 * H2160  # 1 byte per pixel is far above a q70 frame of this content
        self.outs = [gpu.malloc(self.cap) for _ in range(self.F)]
        self.setup_s = time.time() - t0
        self.k = 0
        self.plan = None
        self.units_per_step = self.F * W2160 * H2160

    def step(self):
        # the job array is built once; a step hands every stream its next frame
        if self.plan is None:
            B = self.B
            self.plan = B.JpegBatchPlan(self.gpu, [
                dict(pixels=self.dev_frames[f % self.pool], format=B.BITMAP_FMT_32BIT,
                     width=W2160, height=H2160, stride=W2160 * 4, quality=70, dct=B.DCT_IFAST,
                     out=self.outs[f], out_capacity=self.cap) for f in range(self.F)])
            self.step_pixels = [np.array([self.dev_frames[(f + k) % self.pool] for f in range(self.F)],
                                         dtype=np.uint64) for k in range(self.pool)]
        sizes = self.plan.run(self.step_pixels[self.k % self.pool])
        self.k += 1
        self.last_out = int(sizes.sum())
        return self.last_out

    def config(self, args):
        return {"workload": "mjpeg2160: 3840x2160 BGRX video frames, MJPEG as the reference's stream "
                            "encoder runs libjpeg (IFAST DCT, quality 70, 4:2:0) (BASELINE configs[2])",
                "frames_per_step_per_gpu": self.F, "distinct_frames_per_gpu": self.pool,
                "parallelism": "stream-sharded"}

    def roofline(self, times, steps, out_bytes_timed):
        launches, ms = times.get("jpeg_strip_kernel_ifast", (0, 0.0))
        if not launches:
            return None
        # dominant kernel: pixels in (4 B/px) + Huffman bits out; everything
        # between stays in LDS / registers
        alg = self.F * W2160 * H2160 * 4 + out_bytes_timed / max(1, steps)
        avg_s = ms / launches / 1e3
        ach = alg / avg_s / 1e9
        r = {"bound": "hbm", "kernel": "jpeg_strip_kernel_ifast", "achieved": round(ach, 2),
             "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 5),
             "traffic": pmc_traffic("mjpeg2160", "frames_per_step_per_gpu", self.F),
             "avg_launch_ms": round(ms / launches, 3), "algorithmic_bytes_per_launch": int(alg)}
        # offsets + gather + finish, timed as one group
        tail = times.get("jpeg_bitstream_kernels", (0, 0.0))[1]
        r["bitstream_kernels_ms_per_launch"] = round(tail / launches, 3)
        return r

    def cpu_baseline(self, args):
        """The real libjpeg driven as server/mjpeg_encoder.c drives it (oracle/_ref), else the
        oracle's restatement of it; one thread per stream."""
        from oracle import oracle as O
        threads = max(1, min(16, os.cpu_count() or 1))
        use_ref = O.ref() is not None
        frames_per_thread = 12   # about 20 s of CPU work in all
        px = [f.reshape(H2160, -1) for f in self.host_frames]

        def work(t):
            for k in range(frames_per_thread):
                f = px[(t + k) % len(px)]
                if use_ref:
                    O.ref_jpeg_encode(f, W2160, H2160, W2160 * 4, O.FMT_32BIT, 70, O.DCT_IFAST, 0)
                else:
                    O.jpeg_encode(f, W2160, H2160, W2160 * 4, O.FMT_32BIT, 70, O.DCT_IFAST)

        ths = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
        t0 = time.time()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        dt = time.time() - t0
        return {"value": round(threads * frames_per_thread * W2160 * H2160 / dt / 1e6, 1),
                "unit": "Mpixels/s", "cores": threads, "kind": "reference" if use_ref else "port",
                "sample": "%d threads x %d 3840x2160 frames through %s, %.1f s wall"
                          % (threads, frames_per_thread,
                             "IJG libjpeg 9d called as server/mjpeg_encoder.c:771-823 does"
                             if use_ref else "oracle/jpeg_oracle.c", dt)}


def graduality_line(gpu, mj, args):
    """The codec-choice scorer (_get_bitmap_graduality_level) over the MJPEG workload's frames."""
    from spice_amd import binding as B
    n, reps = 64, 10
    items = [dict(format=B.BITMAP_FMT_32BIT, width=W2160, chunks=[(mj.dev_frames[k % mj.pool], H2160)])
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
            "value": round(n * reps * W2160 * H2160 / dt / 1e6, 1), "unit": "Mpixels/s"}
    if launches:
        ach = alg / (ms / launches / 1e3) / 1e9
        line["roofline"] = {"bound": "hbm", "kernel": "gradual_kernel", "achieved": round(ach, 2),
                            "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 5),
                            "traffic": None, "avg_launch_ms": round(ms / launches, 3),
                            "algorithmic_bytes_per_launch": int(alg)}
    if not args.no_cpu_baseline:
        from oracle import oracle as O
        f = mj.host_frames[0].reshape(H2160, -1)
        t0 = time.time()
        k = 0
        while time.time() - t0 < 2.0:
            O.graduality(O.FMT_32BIT, W2160, [f])
            k += 1
        line["cpu_baseline"] = {"value": round(k * W2160 * H2160 / (time.time() - t0) / 1e6, 1),
                                "unit": "Mpixels/s", "cores": 1, "kind": "port",
                                "sample": "%d 3840x2160 bitmaps through oracle/gradual_oracle.c, one thread" % k}
    return line


def run_workload(cls, gpu, dist, args, steps, warmup):
    wl = cls(gpu, args, dist.rank)
    for _ in range(warmup):
        wl.step()
    gpu.L.spice_gpu_reset_stats()
    gpu.L.spice_gpu_timing_enable(1)
    gpu.L.spice_gpu_timing_reset()
    dist.barrier()
    t0 = time.perf_counter()
    out_bytes = 0
    for _ in range(steps):
        out_bytes += wl.step()
    dist.barrier()
    dt = time.perf_counter() - t0
    gpu.L.spice_gpu_timing_enable(0)
    times = gpu.kernel_times()
    dt_max = dist.max(dt)
    total_units = dist.sum(float(wl.units_per_step * steps))
    return wl, dt_max, total_units, out_bytes, times


def main():
    args = parse_args()
    dist = Dist()
    import spice_amd
    gpu = spice_amd.load(dist.local_rank)

    cls = GlzWorkload if args.workload == "glz1080" else MjpegWorkload
    wl, dt, units, out_bytes, times = run_workload(cls, gpu, dist, args, args.steps, args.warmup)
    codec = 0 if args.workload == "glz1080" else 1
    st = gpu.stats(codec)
    gathered = dist.gather_i64([st["count"], st["orig_bytes"], st["comp_bytes"], st["total_ns"]])

    result = {
        "metric": "compressed Mpixels/s per GPU & 8-GPU node (QUIC/GLZ/MJPEG); % HBM roofline",
        "value": round(units / dt / 1e6, 1),
        "unit": "Mpixels/s",
        "n_gpus": dist.world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": wl.dtype,
        "data": "synthetic",
        "config": wl.config(args),
    }
    if dist.rank == 0:
        result["roofline"] = wl.roofline(times, args.steps, out_bytes)
        result["compression"] = {
            "images": int(sum(g[0] for g in gathered)),
            "orig_MB": round(sum(g[1] for g in gathered) / 1e6, 1),
            "enc_MB": round(sum(g[2] for g in gathered) / 1e6, 1),
        }
        result["kernel_ms"] = {k: round(v[1] / max(1, v[0]), 3) for k, v in times.items() if v[0]}
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = wl.cpu_baseline(args)

    if args.also_mjpeg and args.workload == "glz1080":
        # free the GLZ session state before the second workload
        for e in wl.encs:
            e.close()
        for d in wl.dicts:
            d.close()
        for o in wl.outs:
            gpu.free(o)
        for fr in wl.dev_frames:
            for f in fr:
                gpu.free(f)
        wl2, dt2, units2, out2, times2 = run_workload(MjpegWorkload, gpu, dist, args,
                                                      max(2, args.steps // 2), 1)
        if dist.rank == 0:
            also = {"workload": wl2.config(args)["workload"],
                    "value": round(units2 / dt2 / 1e6, 1), "unit": "Mpixels/s",
                    "roofline": wl2.roofline(times2, max(2, args.steps // 2), out2)}
            if not args.no_cpu_baseline:
                also["cpu_baseline"] = wl2.cpu_baseline(args)
            result["also"] = {"mjpeg2160": also, "graduality2160": graduality_line(gpu, wl2, args)}

    if dist.rank == 0:
        print(json.dumps(result), flush=True)
    dist.close()


if __name__ == "__main__":
    main()
