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
    assert exact and tj["config"] == 3 and "_final_c3.json" in note and 10.0 < tj["bench_avg_launch_us"] < 20.0, note
    tj5, exact5, _ = bench.load_traffic(5, "fast", 100000, 1650.0)
    assert exact5 and tj5["config"] == 5 and tj5["kernels"]["fs2_update"]["hbm_bytes_per_launch"] > 1e9
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
