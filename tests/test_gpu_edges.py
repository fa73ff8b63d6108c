"""Edge cases of the C ABI on the GPU: empty and ragged observation steps, tiny and odd particle counts, capacity and
argument errors (negative return + message, never an abort), history capacity."""
import numpy as np
import pytest

from conftest import sim_args

pytestmark = pytest.mark.gpu
f32 = np.float32
R = np.array([[0.01, 0], [0, 0.017453292519943 ** 2]], f32)
Q = np.array([[0.09, 0], [0, 0.0027415568]], f32)


@pytest.fixture(scope="module")
def sg():
    import slam_amd
    assert slam_amd.device_count() >= 1
    return slam_amd


def empty(n=0):
    return np.zeros((n, 2), f32), np.zeros(n, np.int32)


@pytest.mark.parametrize("N", [1, 2, 63, 65, 257, 1000])
def test_tiny_and_odd_particle_counts(sg, N):
    """Particle counts that are not multiples of the wave or the block (and N = 1, 2): whole Philox-mode runs stay
    finite and normalised, Neff stays in (0, N], estimates and state have the right shapes (parity at ordinary sizes is
    tests/test_gpu_parity.py's business)."""
    from slam_amd import host
    tape = host.make_tape(sim_args("example_webmap", "FASTSLAM2", max(N, 2), 3), max_obs=25)
    s = sg.SlamGpu(N, tape["nlm"], method=2, n_effective=int(0.75 * N), rng_mode=sg.RNG_PHILOX, seed=5, math_mode=0)
    for st in tape["steps"]:
        s.step(np.array(st["controls"], f32).reshape(-1, 3), tape["Q"], float(tape["dt"]), st["zf"], st["idf"], st["zn"], tape["R"])
    est, neff, res = s.history_fetch()
    d = s.download()
    s.close()
    assert est.shape == (25, 3) and np.isfinite(est).all()
    assert d["xv"].shape == (N, 3) and np.isfinite(d["xv"]).all() and np.isfinite(d["w"]).all()
    assert abs(d["w"].sum(dtype=np.float64) - 1.0) < 1e-4
    assert (neff > 0).all() and (neff <= N * (1 + 1e-5)).all()
    if N == 1:
        assert np.allclose(neff, 1.0) and not res.any()  # Neff = 1 >= NEFFECTIVE = 0: a single particle never resamples


def test_step_without_any_observation(sg):
    """m = 0 and n = 0: nothing to update; weights stay uniform, Neff = N, no resample, poses only move by the predicts."""
    N = 500
    s = sg.SlamGpu(N, 8, method=2, n_effective=int(0.75 * N), rng_mode=sg.RNG_PHILOX, seed=1)
    zf, idf = empty()
    for k in range(3):
        s.step(np.array([[3.0, 0.01, 0.0]] * 8, f32), Q, 0.025, zf, idf, zf, R)
    est, neff, res = s.history_fetch()
    d = s.download()
    s.close()
    assert not res.any() and np.allclose(neff, N)
    assert np.allclose(d["w"], 1.0 / N) and d["nf"] == 0
    assert np.allclose(d["xv"], d["xv"][0]) and d["xv"][0, 0] > 1.7  # 24 predicts of 3 m/s x 0.025 s


def test_only_new_landmarks_then_only_reobserved(sg):
    """ragged steps: first only new landmarks (pose sampled from the predicted Gaussian, fastslam2.cpp:36-42), then only
    re-observed ones, then a mix; landmark count and finiteness."""
    N = 300
    s = sg.SlamGpu(N, 8, method=2, n_effective=int(0.75 * N), rng_mode=sg.RNG_PHILOX, seed=2)
    ctl = np.array([[3.0, 0.0, 0.0]] * 8, f32)
    zn = np.array([[10.0, 0.3], [12.0, -0.4], [20.0, 1.0]], f32)
    s.step(ctl, Q, 0.025, *empty(), zn, R)
    assert s.nf() == 3
    zf = np.array([[9.5, 0.31], [19.4, 1.02]], f32)
    s.step(ctl, Q, 0.025, zf, np.array([0, 2], np.int32), np.zeros((0, 2), f32), R)
    s.step(ctl, Q, 0.025, np.array([[10.85, -0.44]], f32), np.array([1], np.int32), np.array([[30.0, 0.0]], f32), R)
    d = s.download()
    est, neff, res = s.history_fetch()
    s.close()
    assert d["nf"] == 4 and np.isfinite(d["xf"]).all() and np.isfinite(d["Pf"]).all() and np.isfinite(d["w"]).all()
    assert est.shape == (3, 3)


def test_argument_and_capacity_errors_are_return_codes(sg):
    N = 64
    s = sg.SlamGpu(N, 2, method=2, n_effective=48, rng_mode=sg.RNG_PHILOX, seed=3)
    ctl = np.array([[3.0, 0.0, 0.0]], f32)
    with pytest.raises(sg.SlamGpuError, match="re-observed"):  # m > landmarks known
        s.step(ctl, Q, 0.025, np.array([[5.0, 0.1]], f32), np.array([0], np.int32), np.zeros((0, 2), f32), R)
    s.step(ctl, Q, 0.025, *empty(), np.array([[5.0, 0.1], [6.0, 0.2]], f32), R)
    with pytest.raises(sg.SlamGpuError, match="capacity"):  # third landmark does not fit max_landmarks = 2
        s.step(ctl, Q, 0.025, *empty(), np.array([[7.0, 0.0]], f32), R)
    with pytest.raises(sg.SlamGpuError, match="out of range"):
        s.step(ctl, Q, 0.025, np.array([[5.0, 0.1]], f32), np.array([5], np.int32), np.zeros((0, 2), f32), R)
    # the context is still usable after the rejected calls
    s.step(ctl, Q, 0.025, np.array([[5.0, 0.1]], f32), np.array([1], np.int32), np.zeros((0, 2), f32), R)
    assert np.isfinite(s.download()["w"]).all()
    s.close()
    with pytest.raises(sg.SlamGpuError):
        sg.SlamGpu(0, 2, method=2)
    with pytest.raises(sg.SlamGpuError, match="EKF1|method"):
        sg.SlamGpu(8, 2, method=0)


def test_history_capacity_is_reported_not_overrun(sg):
    N = 32
    s = sg.SlamGpu(N, 2, method=2, n_effective=24, rng_mode=sg.RNG_PHILOX, seed=4)
    ctl = np.array([[1.0, 0.0, 0.0]], f32)
    zf, idf = empty()
    for k in range(4096):
        s.step(ctl, Q, 0.025, zf, idf, zf, R)
    with pytest.raises(sg.SlamGpuError, match="history full"):
        s.step(ctl, Q, 0.025, zf, idf, zf, R)
    est, _, _ = s.history_fetch()
    assert est.shape == (4096, 3)
    s.step(ctl, Q, 0.025, zf, idf, zf, R)  # usable again after the fetch
    assert s.history_fetch()[0].shape == (1, 3)
    s.close()


@pytest.mark.parametrize("math_mode", [0, 1], ids=["strict", "fast"])
def test_mid_size_map_runs_compact_and_falls_back_to_plain_rows(sg, math_mode, monkeypatch):
    """Round 5: a single context on a map of 40 .. 256 landmarks (example_loop902: 117; fastslam2.cpp:21-48 on that map) uses the
    COMPACT genealogy layout -- 40 rows, four to a 16-byte chunk, packets in the kernel arguments -- like the 35-landmark maps
    do; row consolidation keeps a few-hundred-landmark map in a handful of rows.  It must give, bit for bit, what the plain-row
    layout gives (SLAMGPU_NO_MID_COMPACT=1: the layout of rounds 1-4), over resamples and consolidations; and a step that does
    not fit a kernel-argument packet (here: 45 new landmarks at once) moves the context to plain rows for good, values unchanged."""
    from slam_amd import host
    N = 768
    tape = host.make_tape(sim_args("example_loop902", "FASTSLAM2", N, 3), max_obs=800)
    conf = tape["conf"]
    steps = tape["steps"]
    cap = tape["nlm"] + 50
    rng = np.random.default_rng(5)
    zn_many = np.stack([rng.uniform(5, 20, 45), rng.uniform(-1, 1, 45)], 1).astype(f32)
    outs = []
    for plain in (False, True):
        if plain:
            monkeypatch.setenv("SLAMGPU_NO_MID_COMPACT", "1")
        else:
            monkeypatch.delenv("SLAMGPU_NO_MID_COMPACT", raising=False)
        s = sg.SlamGpu(N, cap, method=2, n_effective=int(0.75 * N), rng_mode=sg.RNG_PHILOX, seed=11, math_mode=math_mode,
                       use_heading=bool(conf.SWITCH_HEADING_KNOWN), wheel_base=float(conf.WHEELBASE), sigma_phi=float(conf.sigmaT))
        in_use0, capacity0 = s.genealogy_rows()
        assert capacity0 == (cap + 1 if plain else 40)
        rows_seen = []
        for k, st in enumerate(steps):
            s.step(np.array(st["controls"], f32).reshape(-1, 3), tape["Q"], float(tape["dt"]), st["zf"], st["idf"], st["zn"], tape["R"])
            if k % 16 == 0:
                rows_seen.append(s.genealogy_rows()[0])
        mid = s.download()
        hist_mid = s.history_fetch()
        # a step no kernel-argument packet holds: 45 new landmarks -> plain rows from here on
        zf0, idf0 = empty()
        s.step(np.zeros((0, 3), f32), tape["Q"], float(tape["dt"]), zf0, idf0, zn_many, tape["R"])
        assert s.genealogy_rows()[1] == cap + 1
        last = steps[-1]
        for _ in range(6):
            s.step(np.array(last["controls"], f32).reshape(-1, 3), tape["Q"], float(tape["dt"]), last["zf"], last["idf"], empty()[0], tape["R"])
        end = s.download()
        hist_end = s.history_fetch()
        outs.append((mid, hist_mid, end, hist_end, max(rows_seen)))
        s.close()
    (ma, ha, ea, ka, rows_c), (mb, hb, eb, kb, rows_p) = outs
    assert ma["nf"] == mb["nf"] and ma["nf"] > 40 and ea["nf"] == eb["nf"] == ma["nf"] + 45
    assert rows_c <= 30 < rows_p, (rows_c, rows_p)    # (consolidation above 6 rows alive; the plain layout lets them pile up)
    for a, b in ((ma, mb), (ea, eb)):
        for key in ("xv", "Pv", "w", "xf", "Pf"):
            assert np.array_equal(a[key].view(np.uint32), b[key].view(np.uint32)), key
    for x, y in zip(ha + ka, hb + kb):
        assert np.array_equal(np.asarray(x), np.asarray(y), equal_nan=True)
    assert np.asarray(ha[2]).sum() > 20   # resamples happened
