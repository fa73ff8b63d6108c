"""`python bench.py --gpus N` starts its own ranks (the driver passes no launcher): the spawn / environment / argument
plumbing of bench.launch_workers at world size 2 on the CPU, with gloo standing in for RCCL."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_launcher_spawns_one_rank_per_gpu_and_rank0_prints_the_line():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "7", "--warmup", "3", "--selftest-plumbing"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["world"] == 2 and j["sum_of_ranks_plus_one"] == 3.0
    assert sorted(map(tuple, j["ranks"])) == [(0, 0, 7, 3, 2), (1, 1, 7, 3, 2)]  # rank, local rank, and the arguments as given


def test_launcher_reports_a_failing_rank():
    import bench
    script = os.path.join(ROOT, "tests", "_rank_exit.py")
    rc = bench.launch_workers(2, ["3"], script=script, timeout=60)
    assert rc == 3
