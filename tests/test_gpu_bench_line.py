"""bench.py end to end on the GPU: ONE JSON line on stdout carrying every field of the driver's contract (plus `roofline`
and `cpu_baseline`), for the default single-GPU run and for the multi-GPU code path launched with one rank."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
            "config", "roofline")


def one_line(out):
    lines = [ln for ln in out.strip().splitlines() if ln.strip()]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def check_common(j, steps, warmup):
    for k in CONTRACT:
        assert k in j, k
    assert j["steps"] == steps and j["warmup"] == warmup and j["n_gpus"] == 1
    assert j["unit"] == "particle-updates/s" and j["higher_is_better"] is True and j["scaling"] == "weak" and j["vs_baseline"] is None
    assert j["dtype"] == "f32" and j["data"] == "synthetic" and "workload" in j["config"]
    assert "FastSLAM2" in j["metric"] and "example_webmap" in j["metric"]
    # value = particles x steps / elapsed, consistent with ms_per_step
    n = j["config"]["particles_total"]
    assert abs(j["value"] * j["ms_per_step"] * 1e-3 / n - 1.0) < 1e-6
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9


def test_default_line_single_gpu():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "60", "--warmup", "5", "--cpu-seconds", "2"], cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = one_line(r.stdout)
    check_common(j, 60, 5)
    assert j["roofline"]["peak"] == 8000.0 and j["roofline"]["kernel"] == "fs2_update"
    assert j["roofline"]["traffic"] and j["roofline"]["avg_launch_us"] > 5.0  # counters of this workload are committed
    assert 1e9 < j["value"] < 2e10
    c = j["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 1e4 and "sample" in c and c["unit"] == j["unit"]
    assert j["config"]["degenerate_steps"] == 0


def test_multi_gpu_path_with_one_rank():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29741", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-sharded", "--steps", "60", "--warmup", "5"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = one_line(r.stdout)
    check_common(j, 60, 5)
    assert j["config"]["multi_gpu_path"] == "dist" and j["config"]["collective"] in ("fold", "push", "rccl")
    # both collectives were tried alone before the run, and the line says which one it took and why
    assert 0.5 < j["config"]["allgather_us"] < 1e3 and 0.5 < j["config"]["flag_barrier_us"] < 1e3
    assert j["config"]["collective_choice"].startswith(j["config"]["collective"])
    chk = j["config"]["check_vs_single_context"]
    assert chk["max_abs_diff"] <= 1e-9, chk  # the distributed run reproduces the single-context estimates
    assert "cpu_baseline" not in j


def test_multi_gpu_path_falls_back_to_the_exchange_path_when_the_peer_mappings_fail():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29743", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0", SLAM_BENCH_FAIL_DIST_SETUP="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-sharded", "--steps", "60", "--warmup", "5"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "falling back to --mgpu exchange" in r.stderr
    j = one_line(r.stdout)
    check_common(j, 60, 5)
    assert "multi_gpu_path" not in j["config"] and "all-to-all" in j["config"]["workload"]
    assert j["config"]["check_vs_single_context"]["max_abs_diff"] <= 1e-9


@pytest.mark.parametrize("collective", ["fold", "push", "native", "torch"])
def test_multi_gpu_path_every_collective(collective):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29745", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-sharded", "--steps", "60", "--warmup", "5",
                        "--collective", collective], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = one_line(r.stdout)
    check_common(j, 60, 5)
    assert j["config"]["collective"] == {"fold": "fold", "push": "push", "native": "rccl", "torch": "torch"}[collective]
    assert j["config"]["check_vs_single_context"]["max_abs_diff"] <= 1e-9
