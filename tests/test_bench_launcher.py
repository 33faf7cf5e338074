"""bench.py's own launcher and line: `--gpus 2` with no launcher in the environment
must start two ranks (gloo here, on the CPU wave-simulator build of the kernels)
and print one JSON line with n_gpus == 2; the other workloads' lines carry the
fields the record needs.  Every run also compares streams with the CPU oracle
after its timed region (bench.py exits non-zero when they differ)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*flags):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--sim", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"] + list(flags), env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_gpus_2_starts_two_ranks(sim):
    d = run_bench("--gpus", "2", "--also", "0")
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1
    assert d["scaling"] == "weak" and d["unit"] == "Mpixels/s" and d["value"] > 0
    assert d["config"]["sessions_per_gpu"] == 3
    assert d["compression"]["images"] == 2 * 3 * 2           # ranks x sessions x timed steps
    assert d["roofline"]["kernel"] in ("glz_parse_kernel", "glz_parse_lone_kernel") and d["verified_sessions"] >= 2


def test_a_launcher_that_disagrees_with_gpus_is_refused():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--sim", "--gpus", "2"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and "--gpus 2" in p.stderr


def test_sessions64_is_sharded_by_session(sim):
    d = run_bench("--gpus", "2", "--workload", "sessions64")
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["glz_sessions_total"] == 32
    assert d["compression"]["images"] == 32 * 2              # every GLZ session once per timed step


def test_multimon_line(sim):
    d = run_bench("--workload", "multimon")
    assert d["config"]["monitors_per_client"] == 4 and d["target_fps"] == 60
    assert d["achieved_fps_per_channel"] > 0 and d["meets_target"] in (True, False)
    assert d["roofline"]["launches_per_step"] == 4           # one launch per display channel
    assert "verified" in d
