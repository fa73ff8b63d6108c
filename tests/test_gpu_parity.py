"""GPU parity tests proper: the HIP path, called through the C ABI (include/slamgpu.h), against
 (a) the golden vectors generated from the reference's own objects (tests/golden/), and
 (b) the oracle on the same seeded inputs (the oracle is bit-pinned to the reference).

Tolerances (float32; the GPU differs from the CPU reference only by device libm, forward substitution
instead of the JacobiSVD solve inside gaussEvaluate, symmetric-packed covariance storage and the
parallel (wave-scan) order of the weight sums):
  pose / landmark means   : 2e-4 absolute (metres / radians) per update from identical pre-state
                            (measured on MI355X: <= 4e-5, median 0 = bit-identical)
  covariances             : 1e-3 relative to the matrix scale
  weights, FastSLAM1      : 1e-3 relative (measured max 1.4e-4)
  weights, FastSLAM2      : median <= 1e-3, 99th percentile <= 5e-2, max <= 0.3 relative, and total-variation
                            distance of the normalised weight vectors <= 2e-2.  FastSLAM2's weight multiplies
                            gaussEvaluate(xv0 - xvs, Pv0) / gaussEvaluate(xv - xvs, Pv) (fastslam2.cpp:360-367) where Pv0,
                            eight predicts after being zeroed, is nearly rank 2 (upstream TODO at fastslam2.cpp:84):
                            its float32 Cholesky has l22 ~ 1e-3, so a 1e-6 m difference in the sampled pose (one ulp
                            of atan2f upstream) moves the weight by ~1e-2.  The reference's own float32 value is
                            up to 26 % away from a float64 evaluation of the same formula (tools note in DESIGN.md),
                            so a tighter bound would test rounding noise, not the kernel.  Measured: median 1.4e-4,
                            p99 1.1e-2, max 7.4e-2.  These are the strict build's bounds; the fast build (restructured
                            arithmetic) has its own, wider weight bounds: see W_TOL below.
  Neff                    : 2e-2 relative (FastSLAM1: 1e-4);  resample decision identical
  ancestors               : follow from the weights: <= 4 % of particles may pick a neighbouring ancestor
                            (FastSLAM1: identical)
"""
import numpy as np
import pytest

from conftest import load_golden, sim_args

pytestmark = pytest.mark.gpu
f32 = np.float32
RM = np.array([[0.1 ** 2, 0], [0, 0.017453292519943 ** 2]], f32)

POSE_ATOL = 2e-4
COV_RTOL = 1e-3


@pytest.fixture(scope="module")
def sg():
    import slam_amd
    assert slam_amd.device_count() >= 1, "GPU tests need a HIP device"
    return slam_amd


def close_cov(a, b, rtol=COV_RTOL):
    scale = max(np.abs(b).max(), 1e-12)
    return np.abs(a.astype(np.float64) - b.astype(np.float64)).max() <= rtol * scale


def sym(P):
    """GPU storage is symmetric-packed (lower triangle); compare against the reference's lower triangle mirrored."""
    L = np.tril(P)
    return L + np.swapaxes(np.tril(P, -1), -1, -2)


# Weight tolerances per build (median, 99th percentile, max of |w_gpu / w_ref - 1|, total-variation distance, share of
# particles whose post-resample ancestor may differ).
#   strict (0): replays the reference's float operations one by one; only the device libm differs.
#   fast   (1): the same step algebraically restructured (device_math.h, fast section).  It cannot replay the
#               reference's roundings, and FastSLAM2's weights amplify rounding: nudging every stored float of the
#               predicted Pv by ONE ulp moves the reference's own weights by median 3e-4, p99 4e-3..9e-3, max 1.6e-2
#               (tests/weight_conditioning.py, CPU, oracle only).  Measured for the fast build over 300 teacher-forced
#               steps: median 1.7e-3 (N=100) / 6.7e-4 (N=1000), p99 3.6e-2, max 1.1e-1, ancestors 3-4 %
#               (tests/parity_report.py); poses and landmarks meet the same absolute tolerances as strict.
#               Against a float64 evaluation of the same update (tests/fs2_float64.py) the float32 REFERENCE is off by
#               median 1.4e-3, p99 2.1e-2, max 6e-2 and the fast build by median 1.9e-3, p99 2.9e-2, max 8.5e-2: the two
#               are equally good float32 evaluations of one formula (test_fast_build_vs_float64_yardstick).  Per step
#               the median can reach 6.5e-3 (a 6-landmark step), hence the per-step bound of 1e-2; a weight error eps
#               moves about N * eps strata across a cumulative-sum boundary, so ONE step at N=100 can swap 9 % of the
#               ancestors (bound 15 %) while the average over the run is 3-4 %.
W_TOL = {0: dict(median=1e-3, p99=5e-2, max=0.16, tv=1e-2, ancestors=0.04),
         1: dict(median=1e-2, p99=1e-1, max=0.25, tv=2e-2, ancestors=0.15)}
# The bounds above hold for EVERY step (a single 6-landmark step has been seen at median 6.5e-3 in the fast build).
# Over a whole run the deviations pool much lower; these aggregate bounds are 2x what tests/parity_report.py measured on
# MI355X (strict: median 1.1e-4, p99 1.05e-2, max 8.1e-2, ancestors 0.7 % at N=100 / 1.5 % at N=1000; fast: median
# 1.7e-3, p99 3.6e-2, max 1.1e-1, ancestors 3.2-3.9 %), so a regression that doubles the error fails.
W_AGG = {0: dict(median=2.5e-4, p99=2.2e-2, max=0.16, ancestors=0.03),
         1: dict(median=4e-3, p99=7e-2, max=0.25, ancestors=0.08)}


def weight_errors(got, exp, math_mode):
    """|w_gpu / w_ref - 1| per particle; the fast build on normalised weights (see compare_weights)"""
    g, e = got.astype(np.float64), exp.astype(np.float64)
    if math_mode == 1:
        g, e = g / g.sum(), e / e.sum()
    return np.abs(g / e - 1.0)


def compare_weights(got, exp, fs2, tag="", math_mode=0):
    rel = np.abs(got.astype(np.float64) / exp.astype(np.float64) - 1.0)
    if not fs2:
        assert rel.max() <= 1e-3, (tag, rel.max())
        return
    tol = W_TOL[math_mode]
    if math_mode == 1:
        # The fast build is compared on NORMALISED weights (what resampleParticles consumes, core.cpp:726) plus a loose
        # check of the common factor: all particles share the same eight predicts from Pv = 0, so their Pv0 are
        # rotations of one matrix and the rounding error of its Cholesky shifts every weight by the same fraction
        # (seen: 0.7 % at a 6-landmark step), which normalisation removes.
        sg_, se_ = got.sum(dtype=np.float64), exp.sum(dtype=np.float64)
        assert abs(sg_ / se_ - 1.0) <= 2e-2, (tag, "common factor", sg_ / se_)
        rel = np.abs((got.astype(np.float64) / sg_) / (exp.astype(np.float64) / se_) - 1.0)
    assert np.median(rel) <= tol["median"] and np.quantile(rel, 0.99) <= tol["p99"] and rel.max() <= tol["max"], \
        (tag, np.median(rel), np.quantile(rel, 0.99), rel.max())
    pg, pe = got.astype(np.float64) / got.sum(dtype=np.float64), exp.astype(np.float64) / exp.sum(dtype=np.float64)
    assert 0.5 * np.abs(pg - pe).sum() <= tol["tv"], (tag, "TV distance", 0.5 * np.abs(pg - pe).sum())


def compare_state(got, exp, fs2=True, tag="", math_mode=0):
    assert got["nf"] == exp["xf"].shape[1], tag
    assert np.abs(got["xv"] - exp["xv"]).max() <= POSE_ATOL, (tag, np.abs(got["xv"] - exp["xv"]).max())
    assert close_cov(got["Pv"], sym(exp["Pv"])), (tag, "Pv")
    compare_weights(got["w"], exp["w"], fs2, tag, math_mode)
    if got["nf"]:
        assert np.abs(got["xf"] - exp["xf"]).max() <= POSE_ATOL * 5, (tag, np.abs(got["xf"] - exp["xf"]).max())
        assert close_cov(got["Pf"], sym(exp["Pf"])), (tag, "Pf")


def test_jacobians_seam1(sg, kat):
    """slamgpu_jacobians in the AcceleratorHandler window layout vs the reference's computeJacobians KAT."""
    n = kat["jac_xv"].shape[0]
    for i in range(0, n, 7):
        zp, Hv, Hf, Sf = sg.jacobians(kat["jac_xv"][i], RM, kat["jac_xf"][i:i + 1], kat["jac_Pf"][i:i + 1])
        np.testing.assert_allclose(zp[0], kat["jac_zp"][i], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(Hv[0], kat["jac_Hv"][i], rtol=2e-6, atol=1e-9)
        np.testing.assert_allclose(Hf[0], kat["jac_Hf"][i], rtol=2e-6, atol=1e-9)
        np.testing.assert_allclose(Sf[0], kat["jac_Sf"][i], rtol=1e-5, atol=1e-9)
    # batched: all features against one pose
    xv = kat["jac_xv"][0]
    zp, Hv, Hf, Sf = sg.jacobians(xv, RM, kat["jac_xf"], kat["jac_Pf"])
    assert zp.shape == (n, 2) and np.isfinite(Sf).all()


@pytest.mark.parametrize("math_mode", [0, 1], ids=["strict", "fast"])
@pytest.mark.parametrize("name,method", [("traj_fs2_webmap_N100_s7", 2), ("traj_fs1_webmap_N100_s7", 1),
                                         ("traj_fs2_webmap_N1000_s1", 2), ("traj_fs2_loop1_N50_s3", 2),
                                         ("traj_fs2_loop2_N100_s7", 2), ("traj_fs1_loop2_N100_s7", 1),
                                         ("traj_fs2_loop902_N100_s3", 2), ("traj_fs1_loop902_N100_s3", 1),
                                         ("traj_fs2_loop902_N1000_s3", 2)])
def test_teacher_forced_update_vs_golden(sg, name, method, math_mode):
    """Upload the reference's pre-update particle set, run ONE slamgpu_update with the reference's tape,
    compare with the reference's post-update particle set.  Every bundled map: example_loop902 (117 landmarks: capacity
    above kSmallRows) is the reference-pinned run of the plain-row kernel, update_kernel<.,0,true>, with the reference's
    linear weights and SWITCH_HEADING_KNOWN (fastslam2.cpp:113-125)."""
    g = load_golden(name)
    cap = max(40, int(g["nf"].max()))
    for k in g["snap_steps"]:
        pre = {key: g["snap%d_pre_%s" % (k, key)] for key in ("xv", "Pv", "w", "xf", "Pf")}
        exp = {key: g["snap%d_post_%s" % (k, key)] for key in ("xv", "Pv", "w", "xf", "Pf")}
        N = pre["w"].shape[0]
        pre["nf"] = pre["xf"].shape[1]
        s = sg.SlamGpu(N, cap, method=method, n_effective=int(g["meta_n_effective"]), use_heading=bool(g["meta_use_heading"]),
                       wheel_base=float(g["meta_wheel_base"]), sigma_phi=float(g["meta_sigma_phi"]), rng_mode=sg.RNG_TAPE,
                       math_mode=math_mode)
        s.upload(pre)
        m, n = g["m"][k - 1], g["n"][k - 1]
        s.update(g["zf"][k - 1, :m], g["idf"][k - 1, :m], g["zn"][k - 1, :n], g["meta_R"], g["snap%d_normals" % k], g["snap%d_sel" % k])
        neff, did, wsum = s.stats()
        assert did == bool(g["resampled"][k - 1]), (name, k)
        np.testing.assert_allclose(neff, g["neff"][k - 1], rtol=2e-2 if method == 2 else 1e-4)
        got = s.download()
        if did:
            # ancestors may differ at cumulative-sum boundaries: compare through the GPU's own ancestor list
            keep = s.ancestors()
            assert np.all(np.diff(keep) >= 0)
            assert np.allclose(got["w"], 1.0 / N)
            # reference ancestors recovered from the oracle-free identity: post particle k == some pre particle;
            # check the bulk of the particles agree with the reference post state
            bad = np.abs(got["xv"] - exp["xv"]).max(axis=1) > POSE_ATOL
            assert bad.mean() <= (W_TOL[math_mode]["ancestors"] if method == 2 else 0.0), (name, k, bad.mean())
            ok = ~bad
            assert np.abs(got["xf"][ok] - exp["xf"][ok]).max() <= POSE_ATOL * 5
        else:
            compare_state(got, exp, fs2=method == 2, tag="%s step %d" % (name, k), math_mode=math_mode)
        s.close()


def test_teacher_forced_predict_vs_golden(sg):
    for name in ("traj_fs2_webmap_N100_s7", "traj_fs2_loop1_N50_s3", "traj_fs2_loop2_N100_s7", "traj_fs2_loop902_N100_s3"):
        g = load_golden(name)
        for c in g["pred_steps"]:
            xv, Pv = g["pred%d_pre_xv" % c], g["pred%d_pre_Pv" % c]
            N = xv.shape[0]
            s = sg.SlamGpu(N, 1, method=2, use_heading=bool(g["meta_use_heading"]), wheel_base=float(g["meta_wheel_base"]),
                           sigma_phi=float(g["meta_sigma_phi"]))
            s.upload(dict(nf=0, xv=xv, Pv=Pv, w=np.full(N, 1.0 / N, f32), xf=None, Pf=None))
            V, G = g["pred%d_VG" % c]
            s.predict(float(V), float(G), g["meta_Q"], float(g["meta_dt"]), float(g["pred%d_phi" % c][0]))
            got = s.download(landmarks=False)
            assert np.abs(got["xv"] - g["pred%d_post_xv" % c]).max() <= 1e-5
            assert close_cov(got["Pv"], sym(g["pred%d_post_Pv" % c]), 1e-4)
            s.close()


def drive_pair(sg, oracle, mapname, method, N, seed, nobs, math_mode=0, per_step=None, want_pre=False, log_weights=False,
               args=None):
    """Run the oracle simulation; feed the GPU context the same controls, observations and RNG tape; at each
    observation step re-synchronise the GPU state to the oracle's (teacher forcing) after comparing.
    log_weights: both sides keep log-weights (slamgpu_config.log_weights / orc_particles_set_log_weights)."""
    from oracle import orc  # noqa: F401  (checker only)
    o = oracle.sim(args if args is not None else sim_args(mapname, method, N, seed))
    if log_weights:
        o.set_log_weights(True)
    algo = o.algo()
    Q, R, dt = o.noise()
    m_id = 2 if method == "FASTSLAM2" else 1
    s = sg.SlamGpu(N, o.nlm, method=m_id, n_effective=algo.n_effective, use_heading=bool(algo.use_heading),
                   add_predict_noise=bool(algo.add_predict_noise), wheel_base=algo.wheel_base, sigma_phi=algo.sigma_phi,
                   rng_mode=sg.RNG_TAPE, math_mode=math_mode, log_weights=log_weights)
    k = 0
    out = []
    while k < nobs:
        a = o.control()
        assert a >= 0
        x, vg = o.true_pose()
        noise2 = None
        if algo.add_predict_noise:
            noise2 = o.last_noise2()
        s.predict(float(vg[0]), float(vg[1]), Q, float(dt), float(x[2]), noise2)
        if a == 1:
            pre = o.particles() if want_pre else None  # predicted, not yet updated
            o.observe()
            ob = o.last_obs()
            normals, sel = o.last_tape()
            s.update(ob["zf"], ob["idf"], ob["zn"], R, normals, sel)
            k += 1
            got = s.download()
            exp = o.particles()
            ne_o, did_o = o.last_resample()
            ne_g, did_g, _ = s.stats()
            out.append(dict(k=k, got=got, exp=exp, neff=(ne_g, ne_o), did=(did_g, did_o), est=(s.estimate(), o.estimate()),
                            keep=s.ancestors() if did_g else None, m=ob["zf"].shape[0], n=ob["zn"].shape[0],
                            pre=pre, obs=ob, normals=normals, R=R))
            if per_step:
                per_step(out[-1])
            s.upload(exp)  # teacher forcing: continue from the oracle's state
    s.close()
    o.close()
    return out


@pytest.mark.parametrize("math_mode", [0, 1], ids=["strict", "fast"])
@pytest.mark.parametrize("method,N,seed,nobs", [("FASTSLAM2", 100, 7, 120), ("FASTSLAM2", 1000, 1, 40),
                                                ("FASTSLAM1", 100, 7, 60), ("FASTSLAM1", 1000, 7, 40),
                                                ("FASTSLAM2", 5000, 12345, 10)])
def test_stepwise_vs_oracle(sg, oracle, method, N, seed, nobs, math_mode):
    """Both kernel builds (strict: no FMA contraction, IEEE divide/sqrt; fast: contraction + the 1-ulp hardware
    v_rcp_f32 / v_sqrt_f32 and the restructured arithmetic) meet the same pose / landmark / covariance tolerances; the
    weight tolerances are per build: W_TOL for every single step, W_AGG (2x the measured figures) over the run.
    FASTSLAM1 N=1000 is BASELINE configs[1] at its real size (the reference resamples for N=1000, core.cpp:751-763)."""
    fs2 = method == "FASTSLAM2"
    rels, anc_bad, anc_tot = [], 0, 0

    def check(r):
        nonlocal anc_bad, anc_tot
        tag = "%s N=%d obs %d (m=%d n=%d)" % (method, N, r["k"], r["m"], r["n"])
        assert r["did"][0] == r["did"][1], tag
        np.testing.assert_allclose(r["neff"][0], r["neff"][1], rtol=2e-2 if fs2 else 1e-4, err_msg=tag)
        if r["did"][0]:
            bad = np.abs(r["got"]["xv"] - r["exp"]["xv"]).max(axis=1) > POSE_ATOL
            # FastSLAM1's weights agree to ~1e-4: identical ancestors at N=100, a boundary stratum or two at N=1000
            # (the fast build's FastSLAM1 update runs the restructured arithmetic since round 4 -- hardware exp / rcp, polynomial
            # atan2: weights to ~1e-5 instead of ~1e-6 -- and one stratum in a few thousand lands on the other side)
            assert bad.mean() <= (W_TOL[math_mode]["ancestors"] if fs2 else ((0.0 if math_mode == 0 else 0.01) if N <= 100 else 0.005)), (tag, bad.mean())
            anc_bad += int(bad.sum())
            anc_tot += bad.size
        else:
            compare_state(r["got"], r["exp"], fs2=fs2, tag=tag, math_mode=math_mode)
            if fs2 and r["m"] > 0:
                rels.append(weight_errors(r["got"]["w"], r["exp"]["w"], math_mode))
        # after a resample a few particles may descend from a neighbouring ancestor (see header): mean moves by <= frac * spread
        np.testing.assert_allclose(r["est"][0][:2], r["est"][1][:2], atol=1e-2 if r["did"][0] else 5e-4, err_msg=tag)
    drive_pair(sg, oracle, "example_webmap", method, N, seed, nobs, math_mode=math_mode, per_step=check)
    if fs2 and rels:
        rel, agg = np.concatenate(rels), W_AGG[math_mode]
        assert np.median(rel) <= agg["median"] and np.quantile(rel, 0.99) <= agg["p99"] and rel.max() <= agg["max"], \
            ("aggregate", np.median(rel), np.quantile(rel, 0.99), rel.max())
        if anc_tot:
            assert anc_bad / anc_tot <= agg["ancestors"], ("aggregate ancestors", anc_bad / anc_tot)


@pytest.mark.parametrize("mapname,seed,nobs", [("example_webmap", 7, 200), ("example_loop2", 7, 200), ("example_loop902", 3, 200)])
def test_fast_build_vs_float64_yardstick(sg, oracle, mapname, seed, nobs):
    """Where the fast build and the float32 reference disagree about a weight, neither is privileged: compare both
    with a float64 evaluation of the same update from the same (float32) predicted particle set.  The fast build's
    error must stay within 2x the reference's own, over all non-resampling steps of the run's first 200 observations, on
    three bundled maps (measured, median / p99 of the fast build's error over the reference's: webmap 1.30 / 1.54, loop2
    1.13 / 1.44, loop902 -- up to ~40 re-observed landmarks per step -- 1.36 / 1.47)."""
    import fs2_float64
    e_gpu, e_ref = [], []

    def collect(r):
        if r["did"][0] or r["did"][1] or r["m"] == 0:
            return
        _, wt = fs2_float64.update_weights(r["pre"], r["obs"]["zf"], r["obs"]["idf"], r["R"], r["normals"])
        wt = wt / wt.sum()
        for dst, w in ((e_gpu, r["got"]["w"]), (e_ref, r["exp"]["w"])):
            w = w.astype(np.float64)
            dst.append(np.abs(w / w.sum() / wt - 1.0))
    drive_pair(sg, oracle, mapname, "FASTSLAM2", 100, seed, nobs, math_mode=1, per_step=collect, want_pre=True)
    e_gpu, e_ref = np.concatenate(e_gpu), np.concatenate(e_ref)
    print("float64 yardstick %s: fast build median %.3g p99 %.3g; float32 reference median %.3g p99 %.3g; ratios %.2f / %.2f over %d weights"
          % (mapname, np.median(e_gpu), np.quantile(e_gpu, 0.99), np.median(e_ref), np.quantile(e_ref, 0.99),
             np.median(e_gpu) / np.median(e_ref), np.quantile(e_gpu, 0.99) / np.quantile(e_ref, 0.99), e_gpu.size))
    assert e_gpu.size >= 2000
    assert np.median(e_gpu) <= 2.0 * np.median(e_ref), (np.median(e_gpu), np.median(e_ref))
    assert np.quantile(e_gpu, 0.99) <= 2.0 * np.quantile(e_ref, 0.99), (np.quantile(e_gpu, 0.99), np.quantile(e_ref, 0.99))
    if mapname == "example_webmap":
        assert np.median(e_ref) >= 3e-4  # the yardstick really is this coarse there: float32 FastSLAM2 weights carry ~1e-3 noise


@pytest.mark.parametrize("math_mode", [0, 1], ids=["strict", "fast"])
def test_free_running_statistics(sg, oracle, math_mode):
    """Free-running (no teacher forcing) GPU run driven by the oracle front end's controls/observations and tape:
    trajectories decorrelate after the first differing ancestor, so this checks the filter's behaviour, not bits:
    the estimated path must track the true path as well as the reference's does."""
    N, seed = 100, 7
    o = oracle.sim(sim_args("example_webmap", "FASTSLAM2", N, seed))
    algo = o.algo()
    Q, R, dt = o.noise()
    s = sg.SlamGpu(N, o.nlm, method=2, n_effective=algo.n_effective, wheel_base=algo.wheel_base, rng_mode=sg.RNG_TAPE,
                   math_mode=math_mode)
    err_g, err_o, first_div = [], [], None
    k = 0
    while k < 400:
        a = o.control()
        x, vg = o.true_pose()
        s.predict(float(vg[0]), float(vg[1]), Q, float(dt), float(x[2]))
        if a == 1:
            o.observe()
            ob = o.last_obs()
            normals, sel = o.last_tape()
            s.update(ob["zf"], ob["idf"], ob["zn"], R, normals, sel)
            k += 1
            eg, eo = s.estimate(), o.estimate()
            err_g.append(np.hypot(eg[0] - x[0], eg[1] - x[1]))
            err_o.append(np.hypot(eo[0] - x[0], eo[1] - x[1]))
            if first_div is None and np.hypot(eg[0] - eo[0], eg[1] - eo[1]) > 1e-3:
                first_div = k
    s.close()
    o.close()
    assert first_div is None or first_div > 3, first_div  # identical (to 1 mm) at least until ancestors first differ
    assert np.mean(err_g) < 1.5 * np.mean(err_o) + 0.05, (np.mean(err_g), np.mean(err_o))


@pytest.mark.parametrize("math_mode", [0, 1], ids=["strict", "fast"])
def test_many_landmarks_big_packet_path(sg, oracle, tmp_path, math_mode):
    """BASELINE config 5 shape in miniature: a synthetic uniform map (400 landmarks on the webmap bounding box,
    MAX_RANGE 60 => dozens of re-observed landmarks per step) drives the device-resident packet path (m, n > 12),
    blockIdx.y > 1 in the resample gather, and landmark capacity growth; checked per step against the oracle."""
    import shutil
    from conftest import DATA
    from slam_amd import host
    import os
    # NB the reference's float32 weight is a product of ~90 per re-observed landmark (1/(2*pi*sqrt(det R))): it overflows
    # to inf (and normalises to NaN) beyond ~20 landmarks per step, so config 5's m ~ 1.3k is outside what the
    # reference arithmetic can represent; this test stays in the representable range (m ~ 14) and treats any
    # non-finite reference weight as "must be non-finite on the GPU too".
    lm = host.synthetic_landmarks(12345, 1000, -130, 100, -100, 90)
    _, wp = host.HostSim(sim_args("example_webmap", "FASTSLAM2", 100, 7)).map()
    mp = str(tmp_path / "syn400.mat")
    host.write_map(mp, lm, wp)
    ini = open(os.path.join(DATA, "example_webmap.ini")).read().replace("MAX_RANGE           = 60.0", "MAX_RANGE           = 20.0")
    open(str(tmp_path / "syn400.ini"), "w").write(ini)
    N = 512
    args = ["-m", mp, "-method", "FASTSLAM2", "-NPARTICLES", N, "-NEFFECTIVE", int(0.75 * N), "-SWITCH_SEED_RANDOM", 3]
    o = oracle.sim(args)
    algo = o.algo()
    Q, R, dt = o.noise()
    s = sg.SlamGpu(N, o.nlm, method=2, n_effective=algo.n_effective, wheel_base=algo.wheel_base, rng_mode=sg.RNG_TAPE,
                   math_mode=math_mode)
    k, max_m = 0, 0
    while k < 10:
        a = o.control()
        x, vg = o.true_pose()
        s.predict(float(vg[0]), float(vg[1]), Q, float(dt), float(x[2]))
        if a == 1:
            o.observe()
            ob = o.last_obs()
            normals, sel = o.last_tape()
            s.update(ob["zf"], ob["idf"], ob["zn"], R, normals, sel)
            k += 1
            max_m = max(max_m, ob["zf"].shape[0], ob["zn"].shape[0])
            got, exp = s.download(), o.particles()
            ne_o, did_o = o.last_resample()
            ne_g, did_g, _ = s.stats()
            assert did_g == did_o and got["nf"] == exp["nf"], k
            if did_g:
                bad = np.abs(got["xv"] - exp["xv"]).max(axis=1) > POSE_ATOL
                assert bad.mean() <= 0.05, (k, bad.mean())
            else:
                assert np.abs(got["xv"] - exp["xv"]).max() <= POSE_ATOL
                assert np.abs(got["xf"] - exp["xf"]).max() <= 2e-3
                fin = np.isfinite(exp["w"]) & (exp["w"] > 0)
                assert np.array_equal(np.isfinite(got["w"]), np.isfinite(exp["w"])), k
                if fin.any():
                    rel = weight_errors(got["w"][fin], exp["w"][fin], math_mode)
                    assert np.median(rel) <= (5e-3 if math_mode == 0 else 2e-2), (k, np.median(rel))
            s.upload(exp)
    assert max_m > 12  # the big-packet path was exercised
    s.close()
    o.close()


@pytest.mark.parametrize("math_mode", [0, 1], ids=["strict", "fast"])
def test_full_size_philox_vs_oracle(sg, oracle, math_mode):
    """BASELINE configs[2] size (100 000 particles, example_webmap) in the throughput RNG mode: the device's
    Philox4x32-10 + Box-Muller stream against the oracle's restatement of the same generator, teacher-forced for
    the first observation steps, plus the size-independent invariants of the update."""
    N = 100000
    o = oracle.sim(sim_args("example_webmap", "FASTSLAM2", N, 7))
    o.set_rng(1, 7)  # particle noise from Philox(seed 7); control / sensor noise from libc rand() as always
    algo = o.algo()
    Q, R, dt = o.noise()
    s = sg.SlamGpu(N, o.nlm, method=2, n_effective=algo.n_effective, wheel_base=algo.wheel_base, rng_mode=sg.RNG_PHILOX, seed=7,
                   math_mode=math_mode)
    # a shadow context that never resamples (NEFFECTIVE 0), stepped in lockstep from the same teacher-forced states with the same
    # Philox keys: its download is the real context's PRE-resample set with the device's own normalised weights
    shadow = sg.SlamGpu(N, o.nlm, method=2, n_effective=0, wheel_base=algo.wheel_base, rng_mode=sg.RNG_PHILOX, seed=7, math_mode=math_mode)
    k = resamples = 0
    while k < 8:   # (the sixth observation step is the first that resamples)
        a = o.control()
        x, vg = o.true_pose()
        s.predict(float(vg[0]), float(vg[1]), Q, float(dt), float(x[2]))
        shadow.predict(float(vg[0]), float(vg[1]), Q, float(dt), float(x[2]))
        if a == 1:
            o.observe_local()
            pre = o.particles()   # the set the resampling stage draws from
            sel = o.last_tape()[1].astype(np.float64)   # the strata of this step (Philox stream 1: the device draws the same)
            own_keep = o.resample()
            ob = o.last_obs()
            s.update(ob["zf"], ob["idf"], ob["zn"], R)
            shadow.update(ob["zf"], ob["idf"], ob["zn"], R)
            w_dev = shadow.download(landmarks=False)["w"].astype(np.float64)
            k += 1
            got, exp = s.download(), o.particles()
            ne_o, did_o = o.last_resample()
            ne_g, did_g, wsum = s.stats()
            assert did_g == did_o, k
            np.testing.assert_allclose(ne_g, ne_o, rtol=2e-2)
            assert np.isfinite(got["w"]).all() and np.isfinite(got["xv"]).all()
            np.testing.assert_allclose(got["w"].sum(dtype=np.float64), 1.0, rtol=1e-4)  # normalised (or N * 1/N)
            if did_g:
                resamples += 1
                keep = s.ancestors()
                assert np.all(np.diff(keep) >= 0) and keep.min() >= 0 and keep.max() < N   # monotone ancestors
                assert np.all(got["w"] == np.float32(1.0) / np.float32(N))
                # every particle is its ancestor's pre-resample self (at this size a large share of the strata pick the
                # neighbour of the oracle's choice -- float32 cumulative sums, tests/test_gpu_freerun.py -- so the comparison
                # goes through the GPU's own ancestors: all particles, no share exempted)
                assert np.abs(got["xv"] - pre["xv"][keep]).max() <= 5e-4, (k, np.abs(got["xv"] - pre["xv"][keep]).max())
                assert np.abs(got["xf"] - pre["xf"][keep]).max() <= POSE_ATOL * 5, k
                # ... and the ancestors themselves, EVERY stratum (round 5; before: "at most 10 % of them more than eight particles
                # from the oracle's").  The oracle's list is drawn from the ORACLE's weights, which differ from the device's by the
                # weight noise of this file's tables, so it cannot pin an index at 10^5 particles; the device's own weights can: with
                # C = the float64 cumulative sum of the device's normalised weights (the shadow context) and t_i the stratum,
                # ancestor a must satisfy C[a-1] <= t_i < C[a] up to what float32 arithmetic can move a cumulative position: the
                # in-block prefix (a scan of 256: <= 8 roundings), the normalisation (2), some slack: 16 * 2^-24 * C[a] -- a tenth of a
                # stratum's width at the far end (core.cpp:800-806 on the device's sums).
                C = np.cumsum(w_dev)
                assert abs(C[-1] - 1.0) < 1e-4
                lo = np.where(keep > 0, C[np.maximum(keep - 1, 0)], 0.0)
                hi = C[keep]
                tol = 16.0 * 2.0 ** -24 * hi + 1e-12
                last = keep == N - 1   # (a stratum beyond the last cumulative weight is clamped to the last particle)
                assert np.all(sel >= lo - tol), (k, int(np.argmax(lo - tol - sel)), float((lo - tol - sel).max()))
                assert np.all((sel < hi + tol) | last), (k, int(np.argmax(sel - hi - tol)), float((sel - hi - tol)[~last].max()))
                # (for the record: how far the oracle's own list, drawn from ITS weights, is from the device's)
                d = np.abs(keep.astype(np.int64) - own_keep.astype(np.int64))
                print("N=1e5 %s step %d: ancestors vs the oracle's own list: identical %.3f, one off %.3f, more than eight off %.4f"
                      % (("strict", "fast")[math_mode], k, (d == 0).mean(), (d == 1).mean(), (d > 8).mean()))
            else:
                # same Philox bits; Box-Muller through device libm (fast build: the hardware v_log / v_sin / v_cos) instead of
                # glibc: poses agree to ~1e-5
                assert np.abs(got["xv"] - exp["xv"]).max() <= 5e-4, (k, np.abs(got["xv"] - exp["xv"]).max())
                rel = weight_errors(got["w"], exp["w"], math_mode)
                assert np.median(rel) <= W_AGG[math_mode]["median"] * 4 and np.quantile(rel, 0.99) <= W_TOL[math_mode]["p99"], \
                    (k, np.median(rel), np.quantile(rel, 0.99))
            np.testing.assert_allclose(s.estimate()[:2], o.estimate()[:2], atol=2e-3)
            s.upload(exp)
            shadow.upload(exp)
    assert resamples >= 1
    s.close()
    shadow.close()
    o.close()


def test_step_call_equals_separate_calls(sg):
    """slamgpu_step (k predicts + update + estimate_async in one C-ABI call) is the same launches as the separate
    calls: states and estimate histories must be bit-identical, across resampling steps (lazy gather included)."""
    import os
    from slam_amd import host
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    N = 4096
    tape = host.make_tape(["-m", os.path.join(root, "data", "example_webmap.mat"), "-method", "FASTSLAM2", "-NPARTICLES", N,
                           "-NEFFECTIVE", int(0.75 * N), "-SWITCH_SEED_RANDOM", 3], max_obs=60)
    runs = []
    for mode in ("separate", "step"):
        s = sg.SlamGpu(N, tape["nlm"], method=2, n_effective=int(0.75 * N), rng_mode=sg.RNG_PHILOX, seed=11, math_mode=1)
        for st in tape["steps"]:
            if mode == "separate":
                for (V, G, phi) in st["controls"]:
                    s.predict(V, G, tape["Q"], float(tape["dt"]), phi)
                s.update(st["zf"], st["idf"], st["zn"], tape["R"])
                s.estimate_async()
            else:
                s.step(np.array(st["controls"], f32).reshape(-1, 3), tape["Q"], float(tape["dt"]), st["zf"], st["idf"], st["zn"], tape["R"])
        est = s.estimate_fetch()
        runs.append((s.download(), est, s.stats()))
        s.close()
    (a, ea, sa), (b, eb, sb) = runs
    assert ea.shape == eb.shape == (len(tape["steps"]), 3) and np.array_equal(ea, eb)
    for key in ("xv", "Pv", "w", "xf", "Pf"):
        assert np.array_equal(a[key], b[key]), key
    assert sa == sb


@pytest.mark.parametrize("method,N,math_mode", [("FASTSLAM1", 1000, 0), ("FASTSLAM1", 1000, 1), ("FASTSLAM1", 70000, 0), ("FASTSLAM1", 70000, 1),
                                                ("FASTSLAM2", 1000, 0)],
                         ids=["fs1-1000-strict", "fs1-1000-fast", "fs1-70000-strict", "fs1-70000-fast", "fs2-1000-strict"])
def test_predict_noise_drawn_at_the_head_of_the_launch_changes_no_bit(sg, method, N, math_mode):
    """Contexts of at most 256 blocks draw the per-particle control noise of the queued predicts (FastSLAM1: always on,
    fastslam1wrapper.cpp:20; FastSLAM2: SWITCH_PREDICT_NOISE) at the head of the update launch, while its first loads are in
    flight (kernels.hip: draw_predict_noise), and FastSLAM1 stages its landmark records like FastSLAM2.  Same Philox counters,
    same Box-Muller: the run must equal, bit for bit, the run whose predicts are flushed as launches of their own
    (predict_kernel draws inside its loop) before every update -- BASELINE configs[1]'s size, a size beyond the 256-block rule,
    and FastSLAM2 with predict noise (strict build only: its fast build runs the generic predict with FMA contraction allowed,
    which the compiler applies differently inside the two kernels; FastSLAM1's fast predict spells its FMAs out)."""
    import os
    from slam_amd import host
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tape = host.make_tape(["-m", os.path.join(root, "data", "example_webmap.mat"), "-method", method, "-NPARTICLES", N,
                           "-NEFFECTIVE", int(0.75 * N), "-SWITCH_SEED_RANDOM", 3], max_obs=80)
    mid = 2 if method == "FASTSLAM2" else 1
    runs = []
    for mode in ("one launch", "predicts flushed"):
        s = sg.SlamGpu(N, tape["nlm"], method=mid, n_effective=int(0.75 * N), add_predict_noise=True, rng_mode=sg.RNG_PHILOX, seed=11,
                       math_mode=math_mode)
        for st in tape["steps"]:
            if mode == "one launch":
                s.step(np.array(st["controls"], f32).reshape(-1, 3), tape["Q"], float(tape["dt"]), st["zf"], st["idf"], st["zn"], tape["R"])
            else:
                for (V, G, phi) in st["controls"]:
                    s.predict(V, G, tape["Q"], float(tape["dt"]), phi)
                s.sync()   # the queued predicts run now, as predict_kernel
                s.update(st["zf"], st["idf"], st["zn"], tape["R"])
                s.estimate_async()
        h = s.history_fetch()
        runs.append((s.download(), h))
        s.close()
    (a, ha), (b, hb) = runs
    assert 5 < ha[2].sum() < 80
    for x, y in zip(ha, hb):
        assert np.array_equal(x, y)
    for key in ("xv", "Pv", "w", "xf", "Pf"):
        assert np.array_equal(a[key].view(np.uint32), b[key].view(np.uint32)), key


def test_history_carries_the_resampling_record(sg):
    """slamgpu_history_fetch: per-step Neff / resampled, identical to what slamgpu_stats reports step by step."""
    import os
    from slam_amd import host
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    N = 2048
    tape = host.make_tape(["-m", os.path.join(root, "data", "example_webmap.mat"), "-method", "FASTSLAM2", "-NPARTICLES", N,
                           "-NEFFECTIVE", int(0.75 * N), "-SWITCH_SEED_RANDOM", 5], max_obs=80)
    logs = []
    for with_stats in (True, False):
        s = sg.SlamGpu(N, tape["nlm"], method=2, n_effective=int(0.75 * N), rng_mode=sg.RNG_PHILOX, seed=2)
        per_step = []
        for st in tape["steps"]:
            s.step(np.array(st["controls"], f32).reshape(-1, 3), tape["Q"], float(tape["dt"]), st["zf"], st["idf"], st["zn"], tape["R"])
            if with_stats:
                ne, rs, _ = s.stats()
                per_step.append((float(ne), bool(rs)))
        xyt, ne, rs = s.history_fetch()
        assert xyt.shape == (len(tape["steps"]), 3)
        if with_stats:
            assert [(float(a), bool(b)) for a, b in zip(ne, rs)] == per_step
            assert 0 < rs.sum() < len(rs)
        logs.append((xyt, ne, rs))
        s.close()
    for a, b in zip(logs[0], logs[1]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("math_mode", [0, 1], ids=["strict", "fast"])
def test_lazy_gather_equals_eager_gather_with_big_packets(sg, tmp_path, math_mode):
    """The lazy gather (update kernel reading through keep[], copy roles skipping the re-observed landmarks by list or,
    for packets that live in device memory, by bitmap) must leave exactly the state the eager gather_kernel leaves:
    run the same Philox-mode steps twice, once reading the state back after every step (forces gather_kernel), once
    without.  Synthetic 1000-landmark map, MAX_RANGE 20: m and n above and below the 12-entry kernel-argument packet,
    several hundred landmarks per particle, resampling on most steps."""
    import os
    from conftest import DATA
    from slam_amd import host
    lm = host.synthetic_landmarks(4321, 1000, -130, 100, -100, 90)
    _, wp = host.HostSim(sim_args("example_webmap", "FASTSLAM2", 100, 7)).map()
    mp = str(tmp_path / "syn1000.mat")
    host.write_map(mp, lm, wp)
    ini = open(os.path.join(DATA, "example_webmap.ini")).read().replace("MAX_RANGE           = 60.0", "MAX_RANGE           = 20.0")
    open(str(tmp_path / "syn1000.ini"), "w").write(ini)
    N = 1024
    tape = host.make_tape(["-m", mp, "-method", "FASTSLAM2", "-NPARTICLES", N, "-NEFFECTIVE", int(0.75 * N), "-SWITCH_SEED_RANDOM", 3], max_obs=60)
    ms = [st["zf"].shape[0] for st in tape["steps"]]
    ns = [st["zn"].shape[0] for st in tape["steps"]]
    assert max(ms) > 12 and min(ms) <= 12 and max(ns) > 8
    out = []
    for eager in (True, False):
        s = sg.SlamGpu(N, tape["nlm"], method=2, n_effective=int(0.75 * N), rng_mode=sg.RNG_PHILOX, seed=5, math_mode=math_mode)
        nres = 0
        for st in tape["steps"]:
            s.step(np.array(st["controls"], f32).reshape(-1, 3), tape["Q"], float(tape["dt"]), st["zf"], st["idf"], st["zn"], tape["R"])
            if eager:
                s.download(landmarks=False)
                nres += int(s.stats()[1])
        if eager:
            assert nres >= 5
        out.append((s.download(), s.estimate_fetch()))
        s.close()
    (a, ea), (b, eb) = out
    assert a["nf"] == b["nf"] and a["nf"] > 40
    finite = np.isfinite(a["w"])
    assert np.array_equal(finite, np.isfinite(b["w"]))
    for key in ("xv", "Pv", "w", "xf", "Pf"):
        assert np.array_equal(a[key].view(np.uint32), b[key].view(np.uint32)), key
    assert np.array_equal(ea, eb, equal_nan=True)


def test_global_scan_path_equals_inline_scan(sg, monkeypatch):
    """Large contexts take the prefix of the block totals from scan_kernel instead of rescanning in every block; same
    function, same association: forcing that path on a small context must not change a single bit."""
    import os
    from slam_amd import host
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    N = 3000
    tape = host.make_tape(["-m", os.path.join(root, "data", "example_webmap.mat"), "-method", "FASTSLAM2", "-NPARTICLES", N,
                           "-NEFFECTIVE", int(0.75 * N), "-SWITCH_SEED_RANDOM", 4], max_obs=60)
    out = []
    for min_blocks in ("1024", "0"):
        monkeypatch.setenv("SLAMGPU_SCAN_MIN_BLOCKS", min_blocks)
        s = sg.SlamGpu(N, tape["nlm"], method=2, n_effective=int(0.75 * N), rng_mode=sg.RNG_PHILOX, seed=6, math_mode=1)
        for st in tape["steps"]:
            s.step(np.array(st["controls"], f32).reshape(-1, 3), tape["Q"], float(tape["dt"]), st["zf"], st["idf"], st["zn"], tape["R"])
        h = s.history_fetch()
        out.append((s.download(), h))
        s.close()
    (a, ha), (b, hb) = out
    assert 5 < ha[2].sum() < 60
    for x, y in zip(ha, hb):
        assert np.array_equal(x, y)
    for key in ("xv", "Pv", "w", "xf", "Pf"):
        assert np.array_equal(a[key].view(np.uint32), b[key].view(np.uint32)), key


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_api_interleavings_do_not_change_results(sg, seed):
    """The step loop defers work (resampling stage planned inside the next launch, estimate reduced two launches later,
    gathers and genealogy kept lazy).  Any observer call between steps (stats, estimate, ancestors, download, sync,
    stand-alone predicts + estimate_async) forces some of it to run early, as launches of its own.  Whatever the
    interleaving, states and histories must equal the undisturbed run bit for bit."""
    import os
    from slam_amd import host
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    N = 1500
    tape = host.make_tape(["-m", os.path.join(root, "data", "example_webmap.mat"), "-method", "FASTSLAM2", "-NPARTICLES", N,
                           "-NEFFECTIVE", int(0.75 * N), "-SWITCH_SEED_RANDOM", 9], max_obs=90)
    Q, R, dt = tape["Q"], tape["R"], float(tape["dt"])

    def run(disturb):
        rng = np.random.default_rng(seed)
        s = sg.SlamGpu(N, tape["nlm"], method=2, n_effective=int(0.75 * N), rng_mode=sg.RNG_PHILOX, seed=3, math_mode=1)
        est_sync = {}
        for k, st in enumerate(tape["steps"]):
            ctl = np.array(st["controls"], f32).reshape(-1, 3)
            if disturb and rng.random() < 0.3:
                # the same step through the separate calls
                for (V, G, phi) in ctl:
                    s.predict(float(V), float(G), Q, dt, float(phi))
                s.update(st["zf"], st["idf"], st["zn"], R)
                s.estimate_async()
            else:
                s.step(ctl, Q, dt, st["zf"], st["idf"], st["zn"], R)
            if disturb:
                for _ in range(rng.integers(0, 3)):
                    what = rng.integers(0, 6)
                    if what == 0:
                        s.stats()
                    elif what == 1:
                        est_sync[k] = s.estimate()
                    elif what == 2:
                        s.ancestors()
                    elif what == 3:
                        s.download(landmarks=bool(rng.integers(0, 2)))
                    elif what == 4:
                        s.sync()
                    else:
                        s.stats()
                        s.estimate()
        h = s.history_fetch()
        d = s.download()
        s.close()
        return d, h, est_sync

    (a, ha, _), (b, hb, es) = run(False), run(True)
    for x, y in zip(ha, hb):
        assert np.array_equal(x, y)
    for key in ("xv", "Pv", "w", "xf", "Pf"):
        assert np.array_equal(a[key].view(np.uint32), b[key].view(np.uint32)), key
    for k, e in es.items():  # the synchronous estimate of step k is the history's entry k
        assert np.array_equal(e, ha[0][k]), k


@pytest.mark.parametrize("math_mode", [0, 1], ids=["strict", "fast"])
def test_queued_predicts_vs_oracle_sequence(sg, oracle, math_mode):
    """k queued predicts (applied by one launch; the fast build folds them into ONE composite step computed on the host
    for a particle at the origin, kernels.h: PredictComposite) against the oracle applying FastSLAM2::predictState k
    times, from random poses and random symmetric positive definite covariances."""
    from oracle import orc
    rng = np.random.default_rng(11)
    N, k = 512, 8
    xv = np.stack([rng.uniform(-100, 100, N), rng.uniform(-100, 100, N), rng.uniform(-3.1, 3.1, N)], 1).astype(f32)
    A = rng.normal(size=(N, 3, 3)) * np.array([0.05, 0.05, 0.01])
    Pv = (A @ A.transpose(0, 2, 1) + np.eye(3) * 1e-6).astype(f32)
    Pv = 0.5 * (Pv + Pv.transpose(0, 2, 1))
    Qm = np.array([[0.09, 0], [0, 0.0027415568]], f32)
    ctl = [(3.0 + 0.1 * j, 0.02 * (j - 3)) for j in range(k)]
    algo = orc.Algo(2, 0, 0, 1, int(0.75 * N), 4.0, 0.0)
    P = oracle.particles(N, 1)
    P.set(dict(nf=0, xv=xv, Pv=Pv, w=np.full(N, 1.0 / N, f32), xf=np.zeros((N, 0, 2), f32), Pf=np.zeros((N, 0, 2, 2), f32)))
    for V, G in ctl:
        P.predict(algo, V, G, Qm, 0.025, 0.0)
    exp = P.get()
    P.close()
    s = sg.SlamGpu(N, 4, method=2, n_effective=int(0.75 * N), wheel_base=4.0, rng_mode=sg.RNG_PHILOX, math_mode=math_mode)
    s.upload(dict(nf=0, xv=xv, Pv=Pv, w=np.full(N, 1.0 / N, f32), xf=None, Pf=None))
    for V, G in ctl:
        s.predict(V, G, Qm, 0.025, 0.0)
    got = s.download(landmarks=False)
    s.close()
    assert np.abs(got["xv"] - exp["xv"]).max() <= 5e-5, np.abs(got["xv"] - exp["xv"]).max()
    assert close_cov(got["Pv"], sym(exp["Pv"]), 2e-4)


def test_row_consolidation_changes_no_bit(sg, monkeypatch):
    """Compact contexts consolidate stale genealogy rows (slamgpu.cpp: do_update: landmarks out of view are rewritten,
    unchanged, into the particles' own slots by an update launch and join the row it opens).  Records move, values do not:
    a whole example_webmap run must be bit for bit the run without consolidation, history included, both methods."""
    import os
    from slam_amd import host
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    N = 2048
    for method, mid in (("FASTSLAM2", 2), ("FASTSLAM1", 1)):
        tape = host.make_tape(["-m", os.path.join(root, "data", "example_webmap.mat"), "-method", method, "-NPARTICLES", N,
                               "-NEFFECTIVE", int(0.75 * N), "-SWITCH_SEED_RANDOM", 4])
        out = []
        for off in (False, True):
            if off:
                monkeypatch.setenv("SLAMGPU_NO_CONSOLIDATE", "1")
            else:
                monkeypatch.delenv("SLAMGPU_NO_CONSOLIDATE", raising=False)
            s = sg.SlamGpu(N, tape["nlm"], method=mid, n_effective=int(0.75 * N), rng_mode=sg.RNG_PHILOX, seed=6, math_mode=1)
            for i, st in enumerate(tape["steps"]):
                s.step(np.array(st["controls"], f32).reshape(-1, 3), tape["Q"], float(tape["dt"]), st["zf"], st["idf"], st["zn"], tape["R"])
                if i == 700:
                    mid_state = s.peek()  # (through the genealogy, mid-run: consolidated or not, the same view)
            h = s.history_fetch()
            out.append((s.download(), h, mid_state))
            s.close()
        monkeypatch.delenv("SLAMGPU_NO_CONSOLIDATE", raising=False)
        (a, ha, ma), (b, hb, mb) = out
        assert len(ha[0]) == len(tape["steps"]) and 500 < ha[2].sum()
        for x, y in zip(ha, hb):
            assert np.array_equal(x, y), method
        for key in ("xv", "Pv", "w", "xf", "Pf"):
            assert np.array_equal(a[key].view(np.uint32), b[key].view(np.uint32)), (method, key)
            assert np.array_equal(ma[key].view(np.uint32), mb[key].view(np.uint32)), (method, key, "mid-run")


def test_plain_row_consolidation_changes_no_bit(sg, tmp_path, monkeypatch):
    """Plain-row contexts (big maps) move the landmarks of their emptiest stale rows into the row each update opens once more
    than a target number of rows is in use (slamgpu.cpp: do_update).  A 1 000-landmark run with the target forced down to 6
    rows (so that every step consolidates) against the same run without: histories, final state, a mid-run view through the
    genealogy: bit for bit; and the rows in use stay bounded where they otherwise grow with every step."""
    import os
    from slam_amd import host
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lmk = host.synthetic_landmarks(4321, 1000, -130, 100, -100, 90)
    h0 = host.HostSim(["-m", os.path.join(root, "data", "example_webmap.mat"), "-method", "FASTSLAM2"])
    _, wp = h0.map()
    h0.close()
    mp = str(tmp_path / "syn1000.mat")
    host.write_map(mp, lmk, wp)
    open(str(tmp_path / "syn1000.ini"), "w").write(open(os.path.join(root, "data", "example_webmap.ini")).read().replace(
        "MAX_RANGE           = 60.0", "MAX_RANGE           = 20.0"))
    N = 1024
    tape = host.make_tape(["-m", mp, "-method", "FASTSLAM2", "-NPARTICLES", N, "-NEFFECTIVE", int(0.75 * N), "-SWITCH_SEED_RANDOM", 3], max_obs=400)
    out = []
    for off in (False, True):
        monkeypatch.setenv("SLAMGPU_PLAIN_ROWS_TARGET", "6")
        if off:
            monkeypatch.setenv("SLAMGPU_NO_CONSOLIDATE", "1")
        else:
            monkeypatch.delenv("SLAMGPU_NO_CONSOLIDATE", raising=False)
        s = sg.SlamGpu(N, tape["nlm"], method=2, n_effective=int(0.75 * N), rng_mode=sg.RNG_PHILOX, seed=6, math_mode=1)
        for i, st in enumerate(tape["steps"]):
            s.step(np.array(st["controls"], f32).reshape(-1, 3), tape["Q"], float(tape["dt"]), st["zf"], st["idf"], st["zn"], tape["R"])
            if i == 250:
                mid_state = s.peek(first=1, stride=3)
        h = s.history_fetch()
        rows = s.live_rows()  # (before the download: it flattens the genealogy)
        out.append((s.download(), h, mid_state, rows))
        s.close()
    monkeypatch.delenv("SLAMGPU_NO_CONSOLIDATE", raising=False)
    monkeypatch.delenv("SLAMGPU_PLAIN_ROWS_TARGET", raising=False)
    (a, ha, ma, ra), (b, hb, mb, rb) = out
    assert len(ha[0]) == len(tape["steps"]) and 50 < ha[2].sum()
    assert ra <= 8 and rb > 40, (ra, rb)
    for x, y in zip(ha, hb):
        assert np.array_equal(x, y)
    for key in ("xv", "Pv", "w", "xf", "Pf"):
        assert np.array_equal(a[key].view(np.uint32), b[key].view(np.uint32)), key
        assert np.array_equal(ma[key].view(np.uint32), mb[key].view(np.uint32)), (key, "mid-run")
