"""MJPEG bit-rate control (server/mjpeg_encoder.c:362-727, 886-1258): the
library's controller against the CPU restatement, event by event, with the
clock and the stream owner's callbacks as test inputs.  The reference holds no
test or fixture for this logic and cannot be compiled here: parity unpinned
(two restatements agree).  The CPU tests use the shipped library's
codec-less controller, which needs no GPU."""
import ctypes

import numpy as np
import pytest

import spice_amd
from spice_amd import binding as B

from spice_amd import frames

from helpers import assert_same_stream

DROP, START = B.MJPEG_ENCODER_FRAME_DROP, B.MJPEG_ENCODER_FRAME_ENCODE_START


@pytest.fixture(scope="module")
def lib():
    return B.SpiceGpu(ctypes.CDLL(spice_amd.LIB_PATH))       # not initialised: no GPU needed


class Scenario:
    """One stream: a content model (frame size per quality), a link and a clock."""

    def __init__(self, seed, bit_rate, src_fps, roundtrip, with_delay_cb=True):
        self.rng = np.random.default_rng(seed)
        self.t = 5_000_000_000 + int(self.rng.integers(0, 10**9))
        self.mm0 = int(self.rng.integers(1, 2**31))
        self.bit_rate = bit_rate
        self.src_fps = src_fps
        self.roundtrip = roundtrip
        self.with_delay_cb = with_delay_cb
        self.base = float(self.rng.integers(1_000, 150_000))   # bytes at quality 50
        self.healthy = seed % 2 == 1           # odd seeds: steady link, no bad news from the client
        self.drop_bias = 0.0 if self.healthy else float(self.rng.choice([0.0, 0.03]))

    def mm(self):
        return (self.mm0 + self.t // 1_000_000) & 0xFFFFFFFF

    def sizes(self):
        r = self.rng
        if r.random() < 0.02:
            self.base = float(r.integers(1_000, 150_000))        # scene change
        self.base = min(max(self.base * float(r.uniform(0.985, 1.015)), 500.0), 2e6)
        gain = [0.45, 0.62, 0.8, 1.0, 1.2, 1.5, 2.0]
        out = [max(1, int(self.base * g * float(r.uniform(0.995, 1.005)))) for g in gain]
        if r.random() < 0.01:
            out[int(r.integers(0, 7))] = 0                      # a frame whose scanlines fail
        return out

    def events(self, n):
        r = self.rng
        for _ in range(n):
            k = r.random()
            if k < 0.88:
                gap = int(r.choice([0, 1, 7, 16, 33, 40, 66, 100, 250, 700, 3500],
                                   p=[.02, .03, .05, .1, .25, .25, .12, .1, .05, .02, .01]))
                self.t += gap * 1_000_000 + int(r.integers(1, 999_999))
                yield ("frame", self.mm(), self.sizes())
            elif k < (0.88 if self.healthy else 0.89 + self.drop_bias):
                self.t += int(r.integers(1, 20_000_000))
                yield ("drop",)
            elif k < 0.985:
                self.t += int(r.integers(1, 50_000_000))
                span = int(r.integers(200, 3000))
                end = self.mm()
                delay = int(r.choice([-400, -40, -16, -15, -1, 0, 5, 40, 120, 400, 2000, 6000]))
                audio = int(r.choice([0xFFFFFFFF, 0, 20, 100, 400, 3000]))
                drops = int(r.choice([0, 0, 0, 1, 5]))
                if self.healthy:                             # a client that keeps up
                    delay, audio, drops = int(r.integers(300, 400)), 0xFFFFFFFF, 0
                yield ("report", int(r.integers(1, 100)), drops, (end - span) & 0xFFFFFFFF, end, delay, audio)
            elif self.healthy:
                self.roundtrip = int(r.choice([10, 20, 40]))
                yield ("env",)
            else:
                self.src_fps = int(r.choice([0, 1, 2, 5, 10, 24, 25, 30, 60]))
                self.roundtrip = int(r.choice([0, 1, 20, 100, 400, 3000]))
                yield ("env",)


def run_pair(lib, oracle, sc, n_events):
    env = B.StreamEnv(sc.roundtrip, sc.src_fps, now_ns=sc.t, with_delay_cb=sc.with_delay_cb)
    enc = B.MJpegEncoder(lib, env=env, starting_bit_rate=sc.bit_rate, rate_only=True)
    ref = oracle.MjpegRate(sc.bit_rate, sc.t, sc.with_delay_cb)
    ref.env(sc.t, sc.roundtrip, sc.src_fps)
    assert enc.rate_state() == ref.state()
    seen = {"frames": 0, "drops": 0, "qualities": set(), "evals": 0, "rate_moves": 0}
    last_rate = ref.bit_rate()
    for step, ev in enumerate(sc.events(n_events)):
        env.now_ns, env.roundtrip_ms, env.source_fps = sc.t, sc.roundtrip, sc.src_fps
        ref.env(sc.t, sc.roundtrip, sc.src_fps)
        if ev[0] == "frame":
            want, want_q = ref.feed(ev[1], ev[2])
            if want < 0:
                break                       # server/mjpeg_encoder.c:627 would have aborted the server
            got, got_q = enc.rate_feed(ev[1], ev[2])
            assert (got, got_q if got == START else 0) == (want, want_q if want == 1 else 0), (step, ev)
            seen["frames" if got == START else "drops"] += 1
            if got == START:
                seen["qualities"].add(got_q)
        elif ev[0] == "drop":
            ref.server_drop()
            enc.notify_server_frame_drop()
        elif ev[0] == "report":
            ref.client_report(*ev[1:])
            enc.client_stream_report(*ev[1:])
        a, b = enc.rate_state(), ref.state()
        assert a == b, (step, ev, {k: (a[k], b[k]) for k in a if a[k] != b[k]})
        assert enc.bit_rate() == ref.bit_rate()
        seen["evals"] += a["during_quality_eval"]
        if ref.bit_rate() != last_rate:
            seen["rate_moves"] += 1
            last_rate = ref.bit_rate()
    if seen["frames"]:
        assert enc.stats()["avg_quality"] == ref.avg_quality()
    assert env.playback_delays[-1:] == ([ref.state()["playback_delay_last"]]
                                        if ref.state()["playback_delay_updates"] else [])
    assert len(env.playback_delays) == ref.state()["playback_delay_updates"]
    enc.close()
    return seen


@pytest.mark.parametrize("seed", range(24))
def test_rate_control_follows_the_restatement(lib, oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    sc = Scenario(seed, bit_rate=int(rng.choice([0, 64_000, 1_000_000, 8_000_000, 50_000_000, 10**9])),
                  src_fps=int(rng.choice([1, 5, 12, 25, 30, 60])),
                  roundtrip=int(rng.choice([0, 10, 80, 600])), with_delay_cb=seed % 5 != 0)
    run_pair(lib, oracle, sc, 3000)


def test_scenarios_reach_every_branch_family(lib, oracle):
    """the generator is only a useful check if streams really probe, settle on
    different qualities, drop frames and move their byte rate"""
    total = {"frames": 0, "drops": 0, "qualities": set(), "evals": 0, "rate_moves": 0}
    for seed in range(100, 112):
        sc = Scenario(seed, bit_rate=[400_000, 4_000_000, 40_000_000][seed % 3], src_fps=25, roundtrip=40)
        seen = run_pair(lib, oracle, sc, 2500)
        for k in total:
            total[k] = total[k] | seen[k] if k == "qualities" else total[k] + seen[k]
    assert total["frames"] > 5000 and total["drops"] > 500
    assert total["qualities"] == {20, 30, 40, 50, 60, 70, 80}
    assert total["evals"] > 200 and total["rate_moves"] > 50


def test_fresh_stream_probes_up_from_the_middle(lib, oracle):
    """a fast link and small frames: quality 50, 60, 70, 80, then settled on 80
    (server/mjpeg_encoder.c:207, 504-522)"""
    env = B.StreamEnv(10, 25, now_ns=10**10)
    enc = B.MJpegEncoder(lib, env=env, starting_bit_rate=100_000_000, rate_only=True)
    sizes = [3000, 4000, 5000, 6000, 7000, 8000, 9000]
    got = []
    for k in range(8):
        env.now_ns += 200_000_000
        r, q = enc.rate_feed(1000 + k * 200, sizes)
        assert r == START
        got.append(q)
    assert got[:5] == [50, 60, 70, 80, 80]
    st = enc.rate_state()
    assert not st["during_quality_eval"] and st["quality_id"] == 6 and st["fps"] == 25
    assert env.playback_delays and env.playback_delays[0] == 5      # 2 * 0 ms send time + 10 / 2
    enc.close()


def test_frames_closer_than_the_frame_interval_are_dropped(lib):
    """pacing starts once the first probe has settled: while a stream probes after a
    rate change no frame time is recorded (server/mjpeg_encoder.c:757-768)"""
    env = B.StreamEnv(0, 25, now_ns=10**10)
    enc = B.MJpegEncoder(lib, env=env, starting_bit_rate=1_600_000, rate_only=True)
    sizes = [20_000] * 7                      # 200 kB/s carry 10 frames a second of these
    for k in range(10):
        env.now_ns += 1_000_000
        assert enc.rate_feed(k, sizes)[0] == START
        if not enc.rate_state()["during_quality_eval"]:
            break
    st = enc.rate_state()
    assert not st["during_quality_eval"] and st["fps"] == 10 and st["adjusted_fps"] == 15.0
    env.now_ns += 200_000_000
    assert enc.rate_feed(20, sizes)[0] == START
    env.now_ns += 50_000_000                 # 15 admitted frames a second: 66.7 ms apart
    assert enc.rate_feed(21, sizes)[0] == DROP
    env.now_ns += 20_000_000
    assert enc.rate_feed(22, sizes)[0] == START
    enc.close()


def test_rate_control_needs_both_callbacks(lib):
    cbs = B.MJpegEncoderRateControlCbs(None, None, None)
    assert not lib.L.spice_gpu_mjpeg_rate_only_new(1000, ctypes.addressof(cbs), None, None, None)
    assert b"get_roundtrip_ms" in lib.L.spice_gpu_last_error()


# ------------------------------------------------------------------ on the GPU
def jfif_quality(jfif):
    """the quality a baseline JFIF was coded with, from its first luminance
    quantiser entry (16 scaled as libjpeg's jpeg_quality_scaling does)"""
    i = bytes(jfif).index(b"\xff\xdb")
    first = jfif[i + 5]
    for q in (20, 30, 40, 50, 60, 70, 80):
        scale = 5000 // q if q < 50 else 200 - 2 * q
        if max(1, min(255, (16 * scale + 50) // 100)) == first:
            return q
    raise AssertionError("unknown quantiser %d" % first)


@pytest.mark.gpu
def test_rate_controlled_stream_on_gpu(gpu, oracle):
    """real frames: the library picks each frame's quality from the sizes its own
    GPU encodes produced; the restatement, fed those sizes, must pick the same,
    and every JFIF must equal the JPEG oracle's at that quality"""
    w, h = 640, 368
    seq = frames.VideoSequence(w, h, seed=3)
    env = B.StreamEnv(30, 25, now_ns=7 * 10**9)
    enc = B.MJpegEncoder(gpu, env=env, starting_bit_rate=6_000_000)
    ref = oracle.MjpegRate(6_000_000, env.now_ns, True)
    plain = B.MJpegEncoder(gpu)              # quality 70, to show the controller really moves it
    coded, dropped, qualities = 0, 0, []
    for k in range(60):
        env.now_ns += int([40, 40, 40, 5, 90][k % 5]) * 1_000_000
        mm = 5000 + k * 40
        frame = seq.next().reshape(h, -1)
        out = enc.encode_frame(B.BITMAP_FMT_32BIT, w, h, frame, w * 4, mm_time=mm)
        ref.env(env.now_ns, env.roundtrip_ms, env.source_fps)
        if out is None:
            dropped += 1
            assert ref.feed(mm, [1] * 7)[0] == 0
            enc.notify_server_frame_drop() if k % 7 == 0 else None
            ref.server_drop() if k % 7 == 0 else None
        else:
            coded += 1
            q = jfif_quality(out)
            qualities.append(q)
            table = [0] * 7
            table[(q - 20) // 10] = len(out)
            assert ref.feed(mm, table) == (1, q)
            want = oracle.jpeg_encode(frame, w, h, w * 4, oracle.FMT_32BIT, q, oracle.DCT_IFAST)
            assert bytes(out) == bytes(want)
        if k % 10 == 9:
            rep = (10, 0, mm - 400, mm, 200, 0xFFFFFFFF)
            enc.client_stream_report(*rep)
            ref.client_report(*rep)
        assert enc.rate_state() == ref.state(), k
    assert coded >= 20 and dropped >= 5 and len(set(qualities)) >= 2
    assert enc.bit_rate() == ref.bit_rate()
    assert jfif_quality(plain.encode_frame(B.BITMAP_FMT_32BIT, w, h, seq.next().reshape(h, -1), w * 4)) == 70
    enc.close()
    plain.close()


def run_stream_agent(dev, oracle):
    """a 200x120 stream cut out of a 320x200 bitmap held in three chunks, top-down and
    bottom-up, with the agent's own 30 fps gate (server/red_worker.c:8341-8377, 8423-8431)"""
    w, h, stride = 320, 200, 320 * 4
    img = frames.golden_image(w, h, "desktop", 5).reshape(h, stride)
    chunks = [img[:70], img[70:71], img[71:]]
    src = (40, 30, 240, 150)
    agent = B.StreamAgent(dev)
    try:
        t = 10**12
        got = agent.marshall_stream_data(B.BITMAP_FMT_32BIT, chunks, h, stride, src, True, 1, t)
        cut = np.ascontiguousarray(img[30:150, 40 * 4:240 * 4])
        assert_same_stream(oracle.jpeg_encode(cut, 200, 120, 800, oracle.FMT_32BIT, 70, oracle.DCT_IFAST),
                           got, "top-down stream frame")
        assert agent.marshall_stream_data(B.BITMAP_FMT_32BIT, chunks, h, stride, src, True, 2,
                                          t + 20_000_000) is None       # sooner than 1/30 s
        got = agent.marshall_stream_data(B.BITMAP_FMT_32BIT, chunks, h, stride, src, False, 3,
                                         t + 60_000_000)
        cut = np.ascontiguousarray(img[50:170, 40 * 4:240 * 4])         # skips y - bottom lines
        assert_same_stream(oracle.jpeg_encode(cut, 200, 120, 800, oracle.FMT_32BIT, 70, oracle.DCT_IFAST),
                           got, "bottom-up stream frame")
        assert (agent.frames_sent, agent.drops_fps) == (2, 1)
        with pytest.raises(B.SpiceGpuError):
            agent.marshall_stream_data(B.BITMAP_FMT_32BIT, chunks[:2], h, stride, src, True, 4, t + 10**9)
    finally:
        agent.close()


def test_sim_stream_agent(sim, oracle):
    run_stream_agent(sim, oracle)


@pytest.mark.gpu
def test_gpu_stream_agent(gpu, oracle):
    run_stream_agent(gpu, oracle)
