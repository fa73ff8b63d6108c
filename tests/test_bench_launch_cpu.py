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


_WARM = []


def _run(cmd, extra_env, timeout=240):
    if not _WARM:
        # the bounds below are seconds: they must outlast the stages BEFORE the one under test, and the first `import torch` of a
        # fresh container takes a minute or two: page it in first
        subprocess.run([sys.executable, "-c", "import torch, torch.distributed"], cwd=ROOT, capture_output=True, timeout=900)
        _WARM.append(1)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "SLAM_BENCH_CRUMBS")}
    env.update(extra_env)
    import time
    t0 = time.time()
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, lines, time.time() - t0


def test_a_rank_that_never_leaves_a_stage_is_named_and_the_launcher_exits_124():
    """the first 8-GPU run must not be wasted on a hang.  Rank 1 of 2 enters the stage `first_collective` and sleeps
    forever (what a rank blocked in ncclCommInitRank / hipIpcOpenMemHandle / the first collective looks like from outside): its
    watchdog -- a child process that touches neither torch nor the GPU -- prints ONE JSON line naming the rank and the last stage it
    reached, every rank is killed, and `python bench.py --gpus 2` exits with 124 well inside the time-out."""
    r, lines, wall = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-plumbing"],
                          {"SLAM_BENCH_TEST_HANG": "1:first_collective", "SLAM_BENCH_STALL_S": "12", "SLAM_BENCH_TIMEOUT_S": "90"})
    assert r.returncode == 124, (r.returncode, r.stdout, r.stderr[-1500:])
    assert len(lines) == 1, r.stdout
    j = lines[0]
    assert j["value"] is None and "stayed in one stage" in j["error"]
    assert j["last_stage"] == "first_collective" and j["n_gpus"] == 2
    # ranks that meet in a collective sit in the same stage, the one that blocks and the one that waits for it: both are suspects, and
    # each one's Python stack (dumped on SIGUSR1 by faulthandler, at C level: GIL or no GIL) says which is which
    assert {x["rank"]: x["stage"] for x in j["ranks"]} == {0: "first_collective", 1: "first_collective"}
    assert sorted(j["suspects"]) == [0, 1] and j["failed_rank"] in (0, 1)
    w = {x["rank"]: " | ".join(x["where"] or []) for x in j["ranks"]}
    assert "in mark" in w[1] and "all_reduce" in w[0], w
    assert 12 <= j["seconds_in_stage"] < 40 and wall < 90, (j, wall)


def test_the_whole_run_is_bounded_too():
    """... and a run whose ranks keep moving from stage to stage but never finish is stopped at SLAM_BENCH_TIMEOUT_S (here a rank
    that hangs at `rendezvous` with the stall bound out of reach)."""
    r, lines, wall = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-plumbing"],
                          {"SLAM_BENCH_TEST_HANG": "0:rendezvous", "SLAM_BENCH_STALL_S": "1000", "SLAM_BENCH_TIMEOUT_S": "20"})
    assert r.returncode == 124 and len(lines) == 1, (r.returncode, r.stdout, r.stderr[-1500:])
    assert "took longer than 20 s" in lines[0]["error"] and lines[0]["last_stage"] == "rendezvous" and 0 in lines[0]["suspects"]
    assert "in mark" in " | ".join(lines[0]["ranks"][0]["where"] or [])
    assert wall < 70


def test_under_torch_distributed_run_a_hanging_rank_fails_the_job_with_the_line():
    """The driver's own launcher for N > 1 (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`): no
    launch_workers in the way, the ranks' watchdogs alone must end the job -- the line on stdout, torchrun non-zero."""
    import bench
    r, lines, wall = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                           "--master-port", str(bench.free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-plumbing"],
                          {"SLAM_BENCH_TEST_HANG": "1:rendezvous", "SLAM_BENCH_STALL_S": "12", "SLAM_BENCH_TIMEOUT_S": "90"})
    assert r.returncode != 0, (r.stdout, r.stderr[-1500:])
    assert len(lines) == 1, r.stdout
    assert 1 in lines[0]["suspects"] and lines[0]["last_stage"] == "rendezvous" and lines[0]["value"] is None
    assert "in mark" in " | ".join(lines[0]["ranks"][1]["where"] or [])
    assert wall < 120


def test_a_rank_that_dies_is_named():
    """a rank that raises (here: rank 1, told to by the test) leaves a `failed: ...` breadcrumb; the line names it, not the ranks the
    launcher terminates a moment later"""
    r, lines, wall = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-plumbing"],
                          {"SLAM_BENCH_TEST_RAISE": "1", "SLAM_BENCH_STALL_S": "30", "SLAM_BENCH_TIMEOUT_S": "60"})
    # (the launcher passes on the rank's own exit code -- or 124 if the watchdog's account was there first)
    assert r.returncode in (1, 124) and len(lines) == 1, (r.returncode, r.stdout, r.stderr[-1500:])
    j = lines[0]
    assert j["failed_rank"] == 1 and j["last_stage"].startswith("failed: RuntimeError") and j["suspects"] == [1], j
    assert wall < 60
