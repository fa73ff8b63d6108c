"""The oracle (oracle/slam_oracle.c) against the golden vectors generated from the reference's own
objects (tests/golden/make_golden.py).  Bit-exact: the oracle restates the reference's float32
operation order (core.cpp / fastslam{1,2}.cpp / vendored Eigen 3.1.3) and is built with the same
x86-64 SSE2 scalar semantics."""
import numpy as np
import pytest

from conftest import bits_equal, load_golden, sim_args

f32 = np.float32
RM = np.array([[0.1 ** 2, 0], [0, 0.017453292519943 ** 2]], f32)
QM = np.array([[0.3 ** 2, 0], [0, 0.052359877559830 ** 2]], f32)


def test_trig_offset(oracle, kat):
    out = np.array([oracle.trig_offset(float(x)) for x in kat["trig_in"]], f32)
    assert bits_equal(out, kat["trig_out"])


def test_compute_jacobians(oracle, kat):
    for i in range(kat["jac_xv"].shape[0]):
        zp, Hv, Hf, Sf = oracle.compute_jacobians(kat["jac_xv"][i], RM, kat["jac_xf"][i:i + 1].copy(), kat["jac_Pf"][i:i + 1].copy())
        assert bits_equal(zp[0], kat["jac_zp"][i])
        assert bits_equal(Hv[0], kat["jac_Hv"][i])
        assert bits_equal(Hf[0], kat["jac_Hf"][i])
        assert bits_equal(Sf[0], kat["jac_Sf"][i])


@pytest.mark.parametrize("D", [2, 3])
def test_gauss_evaluate(oracle, kat, D):
    S, v, out = kat["gauss%d_S" % D], kat["gauss%d_v" % D], kat["gauss%d_out" % D]
    got = np.array([oracle.gauss_evaluate(v[i].copy(), S[i].copy()) for i in range(S.shape[0])], f32)
    fin = np.isfinite(out)
    assert bits_equal(got[fin], out[fin])
    assert np.array_equal(np.isnan(got), np.isnan(out))


def test_cholesky_update(oracle, kat):
    for i in range(kat["chol_x"].shape[0]):
        x, P = oracle.cholesky_update2(kat["chol_x"][i], kat["chol_P"][i], kat["chol_v"][i].copy(), RM, kat["chol_H"][i].copy())
        assert bits_equal(x, kat["chol_xo"][i]) and bits_equal(P, kat["chol_Po"][i])


def test_add_feature(oracle, kat):
    for i in range(kat["addf_xv"].shape[0]):
        xf, Pf = oracle.add_feature(kat["addf_xv"][i].copy(), kat["addf_zn"][i].copy(), RM)
        assert bits_equal(xf, kat["addf_xf"][i]) and bits_equal(Pf, kat["addf_Pf"][i])


@pytest.mark.parametrize("seed", [1, 7, 12345])
def test_rand_tape(oracle, kat, seed):
    oracle.srand(seed)
    assert bits_equal(oracle.randn(2, 1), kat["randn21_%d" % seed])
    oracle.srand(seed)
    assert bits_equal(oracle.randn(3, 1), kat["randn31_%d" % seed])
    oracle.srand(seed)
    assert bits_equal(oracle.randn(1, 9), kat["randn19_%d" % seed])


def test_multivariate_gauss(oracle, kat):
    for i in range(kat["mvg_x"].shape[0]):
        oracle.srand(7 + i)
        g = oracle.randn(3, 1).ravel().copy()
        out = oracle.multivariate_gauss(kat["mvg_x"][i].copy(), kat["mvg_P"][i].copy(), g)
        assert bits_equal(out, kat["mvg_out"][i])


@pytest.mark.parametrize("N", [50, 100, 500, 1000, 5000])
def test_stratified_resample(oracle, kat, N):
    oracle.srand(7)
    cnt, sel = oracle.stratified_random(N)
    assert cnt == N
    keep, neff = oracle.stratified_resample(kat["res%d_w" % N], sel)
    assert np.array_equal(keep, kat["res%d_keep" % N])
    assert bits_equal(np.array([neff]), kat["res%d_neff" % N])
    assert np.all(np.diff(keep) >= 0)


def test_strata_counts(oracle, kat):
    """Which N the reference's stratifiedRandom supports (core.cpp:751-763) — the oracle reports the same counts."""
    for N, c in zip(kat["strata_counts_N"], kat["strata_counts"]):
        oracle.srand(1)
        cnt, sel = oracle.stratified_random(int(N))
        assert cnt == c
        assert np.all(np.diff(sel.astype(np.float64)) > -1e-9) and sel.min() >= 0 and sel.max() <= 1.0


def test_predict_and_heading(oracle, kat):
    for i in range(kat["pred_xv"].shape[0]):
        V, G = kat["pred_VG"][i]
        xv, Pv = oracle.fs2_predict_state(kat["pred_xv"][i], kat["pred_Pv"][i], V, G, QM, 4.0, 0.025)
        assert bits_equal(xv, kat["pred_oxv"][i]) and bits_equal(Pv, kat["pred_oPv"][i])
        hx, hP = oracle.observe_heading(xv, Pv, float(kat["head_phi"][i]), 0.017453292519943)
        assert bits_equal(hx, kat["head_xv"][i]) and bits_equal(hP, kat["head_Pv"][i])
        oracle.srand(100 + i)
        g = oracle.randn(2, 1).ravel().copy()
        x1 = oracle.fs1_predict_state(kat["pred_xv"][i], V, G, QM, 4.0, 0.025, g)
        assert bits_equal(x1, kat["pred1_out"][i])


def test_observe_particle(oracle, kat):
    """sampleProposal + featureUpdate (fastslam2.cpp:28-32) and FS1 computeWeight on single particles."""
    for i in range(kat["obs_xv"].shape[0]):
        xf, Pf = kat["obs_xf"][i].copy(), kat["obs_Pf"][i].copy()
        zf, idf = kat["obs_zf"][i].copy(), kat["obs_idf"][i].copy()
        oracle.srand(1000 + i)
        g = oracle.randn(3, 1).ravel().copy()
        xv, Pv, w = oracle.fs2_sample_proposal(kat["obs_xv"][i], kat["obs_Pv"][i], float(kat["obs_w"][i]), xf, Pf, zf, idf, RM, g)
        xf2, Pf2 = oracle.feature_update(xv, xf, Pf, zf, idf, RM)
        assert bits_equal(xv, kat["obs_o_xv"][i]) and bits_equal(Pv, kat["obs_o_Pv"][i])
        assert bits_equal(np.array([w]), kat["obs_o_w"][i:i + 1])
        assert bits_equal(xf2, kat["obs_o_xf"][i]) and bits_equal(Pf2, kat["obs_o_Pf"][i])
        w1 = oracle.fs1_compute_weight(kat["obs_xv"][i].copy(), xf, Pf, zf, idf, RM)
        assert bits_equal(np.array([w1]), kat["fs1w_out"][i:i + 1])


TRAJ = [("traj_fs2_webmap_N100_s7", "example_webmap", "FASTSLAM2", 100, 7, 2172),
        ("traj_fs1_webmap_N100_s7", "example_webmap", "FASTSLAM1", 100, 7, 2172),
        ("traj_fs2_webmap_N1000_s1", "example_webmap", "FASTSLAM2", 1000, 1, 60),
        ("traj_fs2_webmap_N5000_s12345", "example_webmap", "FASTSLAM2", 5000, 12345, 8),
        ("traj_fs2_loop1_N50_s3", "example_loop1", "FASTSLAM2", 50, 3, 400),
        # round 4: every bundled map has a reference-held trajectory (whole runs; loop902 = 117 landmarks, heading known)
        ("traj_fs2_loop2_N100_s7", "example_loop2", "FASTSLAM2", 100, 7, 1589),
        ("traj_fs1_loop2_N100_s7", "example_loop2", "FASTSLAM1", 100, 7, 1589),
        ("traj_fs2_loop902_N100_s3", "example_loop902", "FASTSLAM2", 100, 3, 4302),
        ("traj_fs1_loop902_N100_s3", "example_loop902", "FASTSLAM1", 100, 3, 4302),
        ("traj_fs2_loop902_N1000_s3", "example_loop902", "FASTSLAM2", 1000, 3, 120)]


@pytest.mark.parametrize("name,mapname,method,N,seed,nobs", TRAJ)
def test_trajectory(oracle, name, mapname, method, N, seed, nobs):
    """Free-running oracle simulation (libc rand() in reference order) against the reference run."""
    g = load_golden(name)
    s = oracle.sim(sim_args(mapname, method, N, seed))
    k = 0
    nctl = 0
    while k < nobs:
        a = s.step()
        assert a >= 0
        nctl += 1
        if a == 1:
            assert nctl == g["ctl"][k]
            ob = s.last_obs()
            m, n = g["m"][k], g["n"][k]
            assert ob["zf"].shape[0] == m and ob["zn"].shape[0] == n
            assert bits_equal(ob["zf"], g["zf"][k, :m]) and np.array_equal(ob["idf"], g["idf"][k, :m])
            assert bits_equal(ob["zn"], g["zn"][k, :n])
            assert s.nf() == g["nf"][k]
            assert np.array_equal(s.estimate(), g["est"][k])
            p = s.particles()
            assert bits_equal(p["w"][:8], g["w_head"][k]) and bits_equal(p["xv"][:8], g["xv_head"][k])
            ne, did = s.last_resample()
            assert did == g["resampled"][k] and bits_equal(np.array([ne]), g["neff"][k:k + 1])
            k += 1
    if nobs == 2172:
        while s.step() >= 0:
            nctl += 1
        assert nctl == 17381  # full webmap run = 17 381 control steps
    s.close()


@pytest.mark.parametrize("name,method", [("traj_fs2_webmap_N100_s7", 2), ("traj_fs1_webmap_N100_s7", 1),
                                         ("traj_fs2_webmap_N1000_s1", 2), ("traj_fs2_loop1_N50_s3", 2),
                                         ("traj_fs2_loop2_N100_s7", 2), ("traj_fs1_loop2_N100_s7", 1),
                                         ("traj_fs2_loop902_N100_s3", 2), ("traj_fs1_loop902_N100_s3", 1),
                                         ("traj_fs2_loop902_N1000_s3", 2)])
def test_teacher_forced_updates(oracle, name, method):
    """orc_update on the reference's pre-update state + tape reproduces the reference's post-update state."""
    from oracle import orc
    g = load_golden(name)
    algo = orc.Algo(method, int(g["meta_use_heading"]), int(g["meta_add_predict_noise"]), int(g["meta_resample"]),
                    int(g["meta_n_effective"]), float(g["meta_wheel_base"]), float(g["meta_sigma_phi"]))
    for k in g["snap_steps"]:
        pre = {key: g["snap%d_pre_%s" % (k, key)] for key in ("xv", "Pv", "w", "xf", "Pf")}
        N = pre["w"].shape[0]
        pre["nf"] = pre["xf"].shape[1]
        P = oracle.particles(N, max(64, pre["nf"] + 8))
        P.set(pre)
        m, n = g["m"][k - 1], g["n"][k - 1]
        keep, neff, did = P.update(algo, g["zf"][k - 1, :m], g["idf"][k - 1, :m], g["zn"][k - 1, :n], g["meta_R"],
                                   np.ascontiguousarray(g["snap%d_normals" % k]), np.ascontiguousarray(g["snap%d_sel" % k]))
        post = P.get()
        for key in ("xv", "Pv", "w", "xf", "Pf"):
            assert bits_equal(post[key], g["snap%d_post_%s" % (k, key)]), (k, key)
        assert did == g["resampled"][k - 1]
        P.close()


def test_teacher_forced_predict(oracle):
    from oracle import orc
    for name in ("traj_fs2_webmap_N100_s7", "traj_fs2_loop1_N50_s3", "traj_fs2_loop2_N100_s7", "traj_fs2_loop902_N100_s3"):
        g = load_golden(name)
        algo = orc.Algo(2, int(g["meta_use_heading"]), 0, 1, int(g["meta_n_effective"]), float(g["meta_wheel_base"]),
                        float(g["meta_sigma_phi"]))
        for c in g["pred_steps"]:
            xv, Pv = g["pred%d_pre_xv" % c], g["pred%d_pre_Pv" % c]
            N = xv.shape[0]
            P = oracle.particles(N, 1)
            P.set(dict(nf=0, xv=xv, Pv=Pv, w=np.full(N, 1.0 / N, f32), xf=np.zeros((N, 0, 2), f32), Pf=np.zeros((N, 0, 2, 2), f32)))
            V, G = g["pred%d_VG" % c]
            P.predict(algo, V, G, g["meta_Q"], float(g["meta_dt"]), float(g["pred%d_phi" % c][0]))
            post = P.get()
            assert bits_equal(post["xv"], g["pred%d_post_xv" % c]) and bits_equal(post["Pv"], g["pred%d_post_Pv" % c])
            P.close()


def test_philox_known_answer(oracle):
    """Philox4x32-10 known-answer vectors (Random123 kat_vectors: zero and pi inputs)."""
    assert [hex(x) for x in oracle.philox((0, 0, 0, 0), (0, 0))] == ['0x6627e8d5', '0xe169c58d', '0xbc57ac4c', '0x9b00dbd8']
    got = oracle.philox((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0))
    assert [hex(x) for x in got] == ['0xd16cfe09', '0x94fdcceb', '0x5001e420', '0x24126ea1']
    got = oracle.philox((0xffffffff,) * 4, (0xffffffff, 0xffffffff))
    assert [hex(x) for x in got] == ['0x408f276d', '0x41c83b0e', '0xa20bc7c6', '0x6d5451fd']


def test_map_reader(oracle):
    import os
    from conftest import DATA
    lm, wp = oracle.read_map(os.path.join(DATA, "example_webmap.mat"))
    assert lm.shape == (2, 35) and wp.shape == (2, 17)
    assert abs(lm[0, 0] - 2.9922) < 1e-6 and abs(lm[1, 0] + 25.7009) < 1e-6


# ---- log-weight extension of the oracle (mirrors slamgpu_config.log_weights) --------------------------------------

@pytest.mark.parametrize("D", [2, 3])
def test_gauss_evaluate_logflag1(oracle, kat, D):
    """The reference's own log branch of gaussEvaluate (fastslam2.cpp:154-160), bit-exact against the reference objects
    (tests/golden/kat_log.npz, generated by make_golden.py log): the building block of the log-weight mode."""
    import os
    from conftest import GOLDEN
    exp = np.load(os.path.join(GOLDEN, "kat_log.npz"))["gauss%d_log" % D]
    S, v = kat["gauss%d_S" % D], kat["gauss%d_v" % D]
    got = np.array([oracle.gauss_evaluate(v[i].copy(), S[i].copy(), 1) for i in range(S.shape[0])], f32)
    fin = np.isfinite(exp)
    assert fin.sum() >= 56 and bits_equal(got[fin], exp[fin])
    # and it IS the log of the linear branch where that one is representable: D = 2 exactly the same normaliser, D = 3
    # differs by the constant sqrt(2 pi) (integer D/2 in the linear branch, fastslam2.cpp:152)
    lin = kat["gauss%d_out" % D]
    ok = fin & np.isfinite(lin) & (lin > 1e-30)
    off = 0.0 if D == 2 else 0.5 * np.log(2 * np.pi)
    assert np.abs(exp[ok] - (np.log(lin[ok].astype(np.float64)) - off)).max() <= 2e-4


@pytest.mark.parametrize("method", ["FASTSLAM2", "FASTSLAM1"])
def test_log_weight_mode_equals_linear_mode_where_representable(oracle, method):
    """example_webmap (at most 7 landmarks per step: the linear float32 weights are representable): the log-weight
    oracle run, fed the same tape, makes the same resampling decisions and exp(log-weight) reproduces the linear
    weights; the poses and maps are identical until ancestors can differ at a cumulative-sum boundary."""
    N, seed, nobs = 100, 7, 150
    runs = []
    for logw in (False, True):
        o = oracle.sim(sim_args("example_webmap", method, N, seed))
        if logw:
            o.set_log_weights(True)
        rec, k = [], 0
        while k < nobs:
            if o.step() == 1:
                k += 1
                rec.append((o.particles(), o.last_resample(), o.estimate()))
        o.close()
        runs.append(rec)
    same_state = True
    nres = 0
    for k, ((pa, (nea, dida), ea), (pb, (neb, didb), eb)) in enumerate(zip(*runs)):
        assert dida == didb, k
        nres += int(dida)
        np.testing.assert_allclose(neb, nea, rtol=2e-3)
        if same_state:
            same_state = np.array_equal(pa["xv"], pb["xv"])
        if same_state:  # identical ancestors so far: everything but the weights' representation is bit-identical
            assert bits_equal(pa["xf"], pb["xf"]) and bits_equal(pa["Pv"], pb["Pv"])
            wl, wb = pa["w"].astype(np.float64), np.exp(pb["w"].astype(np.float64))
            np.testing.assert_allclose(wb / wb.sum(), wl / wl.sum(), rtol=2e-3 if method == "FASTSLAM2" else 1e-4)
            np.testing.assert_allclose(wb.sum(), 1.0, rtol=1e-4)  # normalised log-weights
    assert nres >= 20 and k >= 100


def test_log_weight_mode_survives_many_landmarks(oracle, tmp_path):
    """BASELINE config 5 in miniature: synthetic uniform map, MAX_RANGE 30 => ~80 re-observed landmarks per step: the
    linear float32 weights overflow (inf -> NaN after normalisation), the log-weights stay finite, normalised, and the
    filter keeps tracking the true path."""
    import os
    from conftest import DATA
    from slam_amd import host
    lm = host.synthetic_landmarks(12345, 2000, -130, 100, -100, 90)
    _, wp = host.HostSim(sim_args("example_webmap", "FASTSLAM2", 100, 7)).map()
    mp = str(tmp_path / "syn2000.mat")
    host.write_map(mp, lm, wp)
    open(str(tmp_path / "syn2000.ini"), "w").write(open(os.path.join(DATA, "example_webmap.ini")).read())
    args = ["-m", mp, "-method", "FASTSLAM2", "-NPARTICLES", 64, "-NEFFECTIVE", 48, "-SWITCH_SEED_RANDOM", 3, "-MAX_RANGE", 30]
    out = {}
    for logw in (False, True):
        o = oracle.sim(args)
        if logw:
            o.set_log_weights(True)
        k, ms, errs, finite = 0, [], [], True
        while k < 25:
            if o.step() == 1:
                k += 1
                ms.append(o.last_obs()["zf"].shape[0])
                p = o.particles()
                finite = finite and bool(np.isfinite(p["w"]).all())
                x, _ = o.true_pose()
                e = o.estimate()
                errs.append(np.hypot(e[0] - x[0], e[1] - x[1]))
                if logw and not o.last_resample()[1]:
                    np.testing.assert_allclose(np.exp(p["w"].astype(np.float64)).sum(), 1.0, rtol=1e-3)
        o.close()
        out[logw] = (finite, max(ms), float(np.mean(errs)))
    assert out[True][1] > 40                      # far beyond the ~20 landmarks float32 products survive
    assert not out[False][0]                      # the reference arithmetic really does overflow here
    assert out[True][0] and out[True][2] < 0.5    # log-weights: finite and tracking
