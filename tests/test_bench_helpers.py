"""bench.py's bookkeeping (no GPU): the step window it times, the bytes it charges and the counter file it quotes."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_window_does_not_depend_on_steps_and_warmup():
    # the filter is advanced (untimed) to observation step 1 000, or as late as the run allows: a short run does not
    # measure the first, unrepresentative steps of the map (5 landmarks, resampling every step)
    assert bench.pick_start(2172, 100, 2000, None) == 72
    assert bench.pick_start(2172, 5, 20, None) == 1000
    assert bench.pick_start(2172, 50, 500, None) == 1000
    assert bench.pick_start(2172, 5, 20, 300) == 300
    import pytest
    with pytest.raises(SystemExit):
        bench.pick_start(100, 50, 200, None)


def test_default_window_looks_like_the_run():
    # (round 5) the default window is the first one after step 1 000 whose mean m / n / resample rate are the run's:
    # steps 1005..1025 re-observe 1.65 landmarks per step, the run 3.53; (round 6) to 2 % in m and to the nearest achievable count of
    # resampling steps: at 10^6 particles a step costs 23 + 10.9 m + 46 r microseconds, 10 % of m and r were 7 % of the step
    import numpy as np
    from slam_amd import host
    N = 1000
    tape = host.make_tape(["-m", os.path.join(ROOT, "data", "example_webmap.mat"), "-method", "FASTSLAM1", "-NPARTICLES", N, "-NEFFECTIVE",
                           int(0.75 * N), "-SWITCH_SEED_RANDOM", 7])
    obs = tape["steps"]
    mbar = np.mean([st["zf"].shape[0] for st in obs])
    for warmup, steps in ((5, 20), (20, 200), (100, 2000)):
        s, info = bench.pick_window(obs, warmup, steps, None)
        m = np.mean([st["zf"].shape[0] for st in obs[s + warmup:s + warmup + steps]])
        assert abs(m / mbar - 1) <= 0.02 and abs(info["mean_m"] - m) < 1e-12, (warmup, steps, s, m)
        assert s == bench.pick_window(obs, warmup, steps, None)[0]  # deterministic
        json.dumps(info)
    assert bench.pick_window(obs, 5, 20, None)[0] >= 1000
    # a resample history that never fires in the first candidates pushes the window on; an explicit start is taken as given
    res = np.arange(len(obs) - 100) % 5 < 3   # rate 0.6 ...
    res[1000 - 100:1100 - 100] = False          # ... but for steps 1000..1100
    s_r, info_r = bench.pick_window(obs, 5, 20, None, res, 100)
    lo_r = s_r + 5
    assert not (1000 - 20 < lo_r < 1100) and info_r["resample_rate"] == 0.6 and int(res[lo_r - 100:lo_r - 100 + 20].sum()) == 12, (s_r, info_r)
    assert abs(info_r["mean_m"] / mbar - 1) <= 0.02
    assert bench.pick_window(obs, 5, 20, 1000)[0] == 1000 and bench.pick_window(obs, 5, 20, 1000)[1]["mean_m"] < 2.0


def test_algorithmic_and_design_bytes():
    # B_t = 80 + 40 m + 20 n + r (2 (40 + 20 Nf) + 12)
    assert bench.step_bytes(3, 1, 30, False) == 80 + 120 + 20
    assert bench.step_bytes(3, 1, 30, True) == 80 + 120 + 20 + 2 * (40 + 600) + 12
    # the genealogy layout moves indices instead of records on a resample, and never more than the reference's formulation
    idf = [0, 1, 5]
    quiet = bench.design_bytes(idf, 3, 1, 30, False)
    fired = bench.design_bytes(idf, 3, 1, 30, True)
    assert quiet < fired < bench.step_bytes(3, 1, 31, True)
    assert quiet >= 80 + 40 * 3 + 20


def test_traffic_file_of_the_matching_build_is_quoted():
    tj, exact, note = bench.load_traffic(3, "fast", 100000, 15.0)
    # (the latest round's `final` collection of this workload, whatever the launch time of the box that made it)
    import re
    assert exact and tj["config"] == 3 and re.search(r"traffic_r\d+_(final|mid)_c3\.json", note) and 10.0 < tj["bench_avg_launch_us"] < 20.0, note
    tj5, exact5, _ = bench.load_traffic(5, "fast", 100000, 1650.0)
    assert exact5 and tj5["config"] == 5 and tj5["kernels"]["fs2_update"]["hbm_bytes_per_launch"] > 1e9
    tj4, exact4, note4 = bench.load_traffic(4, "fast", 8 * bench.CONFIGS[4]["particles"], 95.0)  # (what --config 4 --gpus 1 asks for)
    assert exact4 and tj4["config"] == 4 and tj4["kernels"]["fs2_update"]["hbm_bytes_per_launch"] > 1e8, note4   # (update_kernel_wide)
    _, exact_other, note_other = bench.load_traffic(3, "fast", 12345, 16.1)
    assert not exact_other and "NOT this workload" in note_other


def test_every_committed_traffic_file_is_well_formed():
    d = os.path.join(ROOT, "profiles")
    files = [f for f in os.listdir(d) if f.startswith("traffic_") and f.endswith(".json")]
    assert len(files) >= 5
    for f in files:
        tj = json.load(open(os.path.join(d, f)))
        k = next(iter(tj["kernels"].values()))
        assert k["hbm_bytes_per_launch"] > 0 and k["avg_ns_rocprof"] > 0
        assert os.path.exists(os.path.join(ROOT, tj["source"])), (f, tj["source"])


def test_every_tool_a_test_runs_exists():
    """tests start scripts under tools/ as child processes (GPU box only): a pruned tool must not leave a dangling reference"""
    import re
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    missing = []
    for name in sorted(os.listdir(here)):
        if not name.endswith(".py"):
            continue
        text = open(os.path.join(here, name)).read()
        for m in re.finditer(r'"tools",\s*"([A-Za-z0-9_./]+)"|tools/([A-Za-z0-9_/]+\.(?:py|sh))', text):
            rel = m.group(1) or m.group(2)
            if not os.path.exists(os.path.join(root, "tools", rel)):
                missing.append((name, rel))
    assert not missing, missing
