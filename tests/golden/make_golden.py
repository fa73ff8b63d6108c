#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE'S OWN OBJECTS.

Runs only in the authoring container: needs oracle/_ref/libslamref.so, which oracle/Makefile builds from
the sources where they lie under /root/reference (never copied).  The outputs are data only
(inputs + expected outputs); every expected value below comes out of a reference function or a
reference simulation run (oracle/ref_driver.cpp), never out of the oracle.

The RNG tape (normals / strata) stored beside the trajectory snapshots is the one exception: the
reference draws it from libc rand() internally and does not expose it, so it is taken from the
oracle's run of the same seed *after asserting that the oracle's full particle state is bit-identical
to the reference's at that step* (same rand() stream consumed in the same order => same tape).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
f32 = np.float32
KEYS = ["xv", "Pv", "w", "xf", "Pf"]
RM = np.array([[0.1 ** 2, 0], [0, 0.017453292519943 ** 2]], f32)
QM = np.array([[0.3 ** 2, 0], [0, 0.052359877559830 ** 2]], f32)


def make_kats(R):
    rng = np.random.default_rng(20261003)
    d = {}
    # a11 trigonometricOffset sweep
    a = np.concatenate([np.linspace(-20, 20, 161), [np.pi, -np.pi, 2 * np.pi, -2 * np.pi, 7, -7, 0, 1e-8, 6.2831855, -6.2831855]]).astype(f32)
    d["trig_in"] = a
    d["trig_out"] = np.array([R.trig_offset(float(x)) for x in a], f32)
    # a2 computeJacobians
    n = 96
    xv = (rng.normal(size=(n, 3)) * [30, 30, 1.5]).astype(f32)
    xf = (xv[:, :2] + rng.normal(size=(n, 2)) * 25).astype(f32)
    A = rng.normal(size=(n, 2, 2)) * rng.uniform(0.02, 0.5, size=(n, 1, 1))
    Pf = (A @ A.transpose(0, 2, 1) + 1e-4 * np.eye(2)).astype(f32)
    zp = np.zeros((n, 2), f32)
    Hv = np.zeros((n, 2, 3), f32)
    Hf = np.zeros((n, 2, 2), f32)
    Sf = np.zeros((n, 2, 2), f32)
    for i in range(n):
        zp[i], Hv[i], Hf[i], Sf[i] = [x[0] for x in R.compute_jacobians(xv[i], RM, xf[i:i + 1], Pf[i:i + 1])]
    d.update(jac_xv=xv, jac_xf=xf, jac_Pf=Pf, jac_zp=zp, jac_Hv=Hv, jac_Hf=Hf, jac_Sf=Sf)
    # a5 gaussEvaluate D=2,3 incl. near-singular
    for D in (2, 3):
        S = np.zeros((64, D, D), f32)
        v = np.zeros((64, D), f32)
        out = np.zeros(64, f32)
        for i in range(64):
            B = rng.normal(size=(D, D)) * rng.uniform(0.005, 1)
            eps = 1e-4 if i % 4 else 1e-9
            S[i] = (B @ B.T + eps * np.eye(D)).astype(f32)
            if i % 8 == 7:  # nearly rank-deficient, like Pv after a few predicts
                u = rng.normal(size=(D, 1))
                S[i] = (u @ u.T * 0.01 + 1e-7 * np.eye(D)).astype(f32)
            v[i] = (rng.normal(size=D) * 0.05).astype(f32)
            out[i] = R.gauss_evaluate(v[i], S[i])
        d["gauss%d_S" % D], d["gauss%d_v" % D], d["gauss%d_out" % D] = S, v, out
    # a8 choleskyUpdate (2x2)
    n = 64
    x = (rng.normal(size=(n, 2)) * 20).astype(f32)
    B = rng.normal(size=(n, 2, 2)) * 0.3
    P = (B @ B.transpose(0, 2, 1) + 1e-3 * np.eye(2)).astype(f32)
    v = (rng.normal(size=(n, 2)) * [0.1, 0.02]).astype(f32)
    H = rng.normal(size=(n, 2, 2)).astype(f32)
    xo = np.zeros_like(x)
    Po = np.zeros_like(P)
    for i in range(n):
        xo[i], Po[i] = R.cholesky_update2(x[i], P[i], v[i], RM, H[i])
    d.update(chol_x=x, chol_P=P, chol_v=v, chol_H=H, chol_xo=xo, chol_Po=Po)
    # a9 addFeature
    xv = (rng.normal(size=(32, 3)) * [30, 30, 1.5]).astype(f32)
    zn = np.stack([rng.uniform(1, 60, size=(32, 3)), rng.uniform(-1.5, 1.5, size=(32, 3))], -1).astype(f32)
    axf = np.zeros((32, 3, 2), f32)
    aPf = np.zeros((32, 3, 2, 2), f32)
    for i in range(32):
        axf[i], aPf[i] = R.add_feature(xv[i], zn[i], RM)
    d.update(addf_xv=xv, addf_zn=zn, addf_xf=axf, addf_Pf=aPf)
    # a6 rand stream heads + randn + multivariateGauss
    for seed in (1, 7, 12345):
        d["rand_%d" % seed] = R.rand_stream(seed, 32)
        d["randn21_%d" % seed] = R.randn(seed, 2, 1)
        d["randn31_%d" % seed] = R.randn(seed, 3, 1)
        d["randn19_%d" % seed] = R.randn(seed, 1, 9)
    B = rng.normal(size=(16, 3, 3)) * 0.2
    P3 = (B @ B.transpose(0, 2, 1) + 1e-4 * np.eye(3)).astype(f32)
    x3 = (rng.normal(size=(16, 3)) * 5).astype(f32)
    d["mvg_P"], d["mvg_x"] = P3, x3
    d["mvg_out"] = np.stack([R.multivariate_gauss(7 + i, x3[i], P3[i]) for i in range(16)])
    # a10 stratified resample
    for N in (50, 100, 500, 1000, 5000):
        w = rng.uniform(0.0, 1.0, size=N).astype(f32) ** 3
        keep, neff = R.stratified_resample(7, w)
        d["res%d_w" % N], d["res%d_keep" % N], d["res%d_neff" % N] = w, keep, np.array([neff], f32)
    d["strata_counts_N"] = np.array([10, 30, 50, 64, 100, 128, 500, 1000, 1024, 2000, 5000, 10000, 100000, 1000000], np.int64)
    d["strata_counts"] = np.array([R.stratified_count(int(N)) for N in d["strata_counts_N"]], np.int64)
    # a12 predictState (FS2, no noise), a13 observeHeading, a15 FS1 predictState with libc noise
    n = 48
    xv = (rng.normal(size=(n, 3)) * [30, 30, 1.5]).astype(f32)
    B = rng.normal(size=(n, 3, 3)) * 0.05
    Pv = (B @ B.transpose(0, 2, 1)).astype(f32)
    Pv[::4] = 0
    VG = np.stack([rng.normal(3, 0.3, n), rng.normal(0, 0.3, n)], -1).astype(f32)
    oxv = np.zeros_like(xv)
    oPv = np.zeros_like(Pv)
    hxv = np.zeros_like(xv)
    hPv = np.zeros_like(Pv)
    f1 = np.zeros_like(xv)
    phi = (xv[:, 2] + rng.normal(0, 0.02, n)).astype(f32)
    for i in range(n):
        oxv[i], oPv[i] = R.fs2_predict_state(xv[i], Pv[i], VG[i, 0], VG[i, 1], QM, 4.0, 0.025)
        hxv[i], hPv[i] = R.observe_heading(oxv[i], oPv[i], float(phi[i]), 0.017453292519943)
        f1[i] = R.fs1_predict_state(100 + i, xv[i], VG[i, 0], VG[i, 1], QM, 4.0, 0.025)
    d.update(pred_xv=xv, pred_Pv=Pv, pred_VG=VG, pred_oxv=oxv, pred_oPv=oPv, head_phi=phi, head_xv=hxv, head_Pv=hPv, pred1_out=f1)
    # a3+a4+a7 on one particle / a14 FS1 weight
    n = 64
    nf = 5
    xv = (rng.normal(size=(n, 3)) * [30, 30, 1.5]).astype(f32)
    B = rng.normal(size=(n, 3, 3)) * [[0.05], [0.05], [0.01]]
    Pv = (B @ B.transpose(0, 2, 1) + 1e-8 * np.eye(3)).astype(f32)
    xf = (xv[:, None, :2] + rng.normal(size=(n, nf, 2)) * 25).astype(f32)
    B = rng.normal(size=(n, nf, 2, 2)) * 0.1
    Pf = (B @ B.transpose(0, 1, 3, 2) + 1e-4 * np.eye(2)).astype(f32)
    m = 3
    idf = np.stack([rng.permutation(nf)[:m] for _ in range(n)]).astype(np.int32)
    zf = np.zeros((n, m, 2), f32)
    w_in = rng.uniform(0.001, 0.02, n).astype(f32)
    o = dict(xv=np.zeros_like(xv), Pv=np.zeros_like(Pv), w=np.zeros(n, f32), xf=np.zeros_like(xf), Pf=np.zeros_like(Pf))
    w1 = np.zeros(n, f32)
    for i in range(n):
        zpi = R.compute_jacobians(xv[i], RM, np.ascontiguousarray(xf[i][idf[i]]), np.ascontiguousarray(Pf[i][idf[i]]))[0]
        zf[i] = (zpi + rng.normal(size=(m, 2)) * [0.1, 0.0175]).astype(f32)
        o["xv"][i], o["Pv"][i], o["w"][i], o["xf"][i], o["Pf"][i] = R.fs2_observe_particle(
            1000 + i, xv[i], Pv[i], float(w_in[i]), xf[i], Pf[i], zf[i], idf[i], RM)
        w1[i] = R.fs1_compute_weight(xv[i], xf[i], Pf[i], zf[i], idf[i], RM)
    d.update(obs_xv=xv, obs_Pv=Pv, obs_w=w_in, obs_xf=xf, obs_Pf=Pf, obs_idf=idf, obs_zf=zf, fs1w_out=w1,
             **{"obs_o_" + k: v for k, v in o.items()})
    np.savez_compressed(os.path.join(OUT, "kat_functions.npz"), **d)
    print("kat_functions.npz:", len(d), "arrays")


def run_pair(R, O, args, snap_steps, max_obs, pred_snaps=()):
    """Reference run (truth) + oracle run (tape), asserting bit equality at every observation step."""
    # pass 1: the reference
    r = R.sim(args)
    ref = []
    snaps = {}
    preds = {}
    nctl = 0
    while True:
        before = r.particles() if (nctl + 1) in pred_snaps else None
        a = r.control()
        if a < 0:
            break
        nctl += 1
        if before is not None:
            x, vg = r.true_pose()
            preds[nctl] = dict(pre=before, post=r.particles(), VG=vg, phi=x[2])
        if a == 1:
            k = len(ref) + 1
            pre = r.particles() if k in snap_steps else None
            r.observe()
            post = r.particles()
            ob = r.last_obs()
            ref.append(dict(ctl=nctl, post=post, obs=ob, est=r.estimate(), true=r.true_pose()[0]))
            if pre is not None:
                snaps[k] = dict(pre=pre, post=post, obs=ob)
            if len(ref) >= max_obs:
                break
    r.close()
    # pass 2: the oracle, same seed => same rand() stream
    o = O.sim(args)
    algo = o.algo()
    k = 0
    nctl = 0
    while True:
        a = o.control()
        if a < 0:
            break
        nctl += 1
        if nctl in preds:
            po = o.particles()
            for key in KEYS:
                assert np.array_equal(po[key].view(np.uint32), preds[nctl]["post"][key].view(np.uint32)), ("pred", nctl, key)
        if a == 1:
            o.observe()
            k += 1
            po = o.particles()
            for key in KEYS:
                assert np.array_equal(po[key].view(np.uint32), ref[k - 1]["post"][key].view(np.uint32)), (k, key)
            ne, did = o.last_resample()
            ref[k - 1]["neff"], ref[k - 1]["resampled"] = ne, did
            if k in snaps:
                snaps[k]["normals"], snaps[k]["sel"] = o.last_tape()
            if k >= max_obs:
                break
    Q, Rn, dt = o.noise()
    meta = dict(Q=Q, R=Rn, dt=dt, n_effective=algo.n_effective, wheel_base=algo.wheel_base, sigma_phi=algo.sigma_phi,
                use_heading=algo.use_heading, add_predict_noise=algo.add_predict_noise, resample=algo.resample)
    o.close()
    return ref, snaps, preds, meta


def save_traj(name, ref, snaps, preds, meta, nshow=8):
    d = {}
    T = len(ref)
    d["ctl"] = np.array([r["ctl"] for r in ref], np.int32)
    d["m"] = np.array([r["obs"]["zf"].shape[0] for r in ref], np.int32)
    d["n"] = np.array([r["obs"]["zn"].shape[0] for r in ref], np.int32)
    d["nf"] = np.array([r["post"]["nf"] for r in ref], np.int32)
    d["est"] = np.stack([r["est"] for r in ref])
    d["true"] = np.stack([r["true"] for r in ref])
    d["neff"] = np.array([r["neff"] for r in ref], f32)
    d["resampled"] = np.array([r["resampled"] for r in ref], np.bool_)
    d["w_head"] = np.stack([r["post"]["w"][:nshow] for r in ref])
    d["xv_head"] = np.stack([r["post"]["xv"][:nshow] for r in ref])
    d["w_sum"] = np.array([r["post"]["w"].astype(np.float64).sum() for r in ref])
    # the observation tape (RNG dependent: sensor noise) so that a driver can be fed draw-for-draw
    mm = max(1, int(d["m"].max()))
    nn = max(1, int(d["n"].max()))
    zf = np.zeros((T, mm, 2), f32)
    idf = np.full((T, mm), -1, np.int32)
    zn = np.zeros((T, nn, 2), f32)
    for i, r in enumerate(ref):
        zf[i, :d["m"][i]] = r["obs"]["zf"]
        idf[i, :d["m"][i]] = r["obs"]["idf"]
        zn[i, :d["n"][i]] = r["obs"]["zn"]
    d.update(zf=zf, idf=idf, zn=zn)
    for k, v in meta.items():
        d["meta_" + k] = np.asarray(v)
    for k, s in snaps.items():
        for key in KEYS:
            d["snap%d_pre_%s" % (k, key)] = s["pre"][key]
            d["snap%d_post_%s" % (k, key)] = s["post"][key]
        d["snap%d_normals" % k] = s["normals"]
        d["snap%d_sel" % k] = s["sel"]
    d["snap_steps"] = np.array(sorted(snaps), np.int32)
    for c, s in preds.items():
        for key in ("xv", "Pv"):
            d["pred%d_pre_%s" % (c, key)] = s["pre"][key]
            d["pred%d_post_%s" % (c, key)] = s["post"][key]
        d["pred%d_VG" % c] = s["VG"]
        d["pred%d_phi" % c] = np.array([s["phi"]], f32)
    d["pred_steps"] = np.array(sorted(preds), np.int32)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **d)
    print(name, "obs steps", T, "snaps", sorted(snaps), "preds", sorted(preds), "%.0f KB" % (os.path.getsize(path) / 1024))


def main():
    orc.build_ref()
    orc.build_oracle()
    R = orc.Reference()
    O = orc.Oracle()
    make_kats(R)
    data = os.path.join(ROOT, "data")

    def args(mapname, method, N, seed):
        return ["-m", os.path.join(data, mapname + ".mat"), "-method", method, "-NPARTICLES", N,
                "-NEFFECTIVE", int(0.75 * N), "-SWITCH_SEED_RANDOM", seed]

    # FS2 webmap N=100 seed 7: whole run (2172 observation steps) + teacher-forcing snapshots
    a = args("example_webmap", "FASTSLAM2", 100, 7)
    ref, _, _, _ = run_pair(R, O, a, set(), 10 ** 9)
    m = np.array([r["obs"]["zf"].shape[0] for r in ref])
    n = np.array([r["obs"]["zn"].shape[0] for r in ref])
    res = np.array([r["resampled"] for r in ref])
    steps = {1, 2, 3, int(np.argmax(m)) + 1, len(ref)}
    both = np.where((m > 0) & (n > 0))[0]
    if len(both):
        steps.add(int(both[0]) + 1)
    steps.add(int(np.where(res)[0][0]) + 1)
    steps.add(int(np.where(~res & (m > 0))[0][5]) + 1)
    steps.add(int(np.where(res & (m >= 5))[0][-1]) + 1)
    ref, snaps, preds, meta = run_pair(R, O, a, steps, 10 ** 9, pred_snaps={1, 2, 9, 10, 500})
    save_traj("traj_fs2_webmap_N100_s7", ref, snaps, preds, meta)
    # FS1 webmap N=100 seed 7: whole run + snapshots
    a = args("example_webmap", "FASTSLAM1", 100, 7)
    ref, snaps, preds, meta = run_pair(R, O, a, {1, 2, 3, 40, 700}, 10 ** 9, pred_snaps={1, 9, 300})
    save_traj("traj_fs1_webmap_N100_s7", ref, snaps, preds, meta)
    # larger N, other seeds: summaries only (+1 snapshot)
    a = args("example_webmap", "FASTSLAM2", 1000, 1)
    ref, snaps, preds, meta = run_pair(R, O, a, {25}, 300)
    save_traj("traj_fs2_webmap_N1000_s1", ref, snaps, preds, meta)
    a = args("example_webmap", "FASTSLAM2", 5000, 12345)
    ref, snaps, preds, meta = run_pair(R, O, a, set(), 60)
    save_traj("traj_fs2_webmap_N5000_s12345", ref, snaps, preds, meta)
    # heading-known map (josephUpdate path, a13)
    a = args("example_loop1", "FASTSLAM2", 50, 3)
    ref, snaps, preds, meta = run_pair(R, O, a, {2, 30}, 400, pred_snaps={1, 2, 30})
    save_traj("traj_fs2_loop1_N50_s3", ref, snaps, preds, meta)
    make_ekf(R, "traj_ekf_loop1_s3", "example_loop1", 3)
    make_ekf(R, "traj_ekf_webmap_s7", "example_webmap", 7)


def main_maps():
    """Round 4: the two bundled maps that had no reference-held fixture.  example_loop2 (25 landmarks, heading known,
    compact genealogy) and example_loop902 (117 landmarks, heading known, MAX_RANGE 10: the only reference-runnable
    workload on the plain-row kernel with the reference's linear weights).  Same protocol as main(): reference objects =
    truth, oracle asserted bit-identical at every observation step, tape taken from the oracle."""
    orc.build_ref()
    orc.build_oracle()
    R = orc.Reference()
    O = orc.Oracle()
    data = os.path.join(ROOT, "data")

    def args(mapname, method, N, seed):
        return ["-m", os.path.join(data, mapname + ".mat"), "-method", method, "-NPARTICLES", N,
                "-NEFFECTIVE", int(0.75 * N), "-SWITCH_SEED_RANDOM", seed]

    def pick(ref, extra=()):
        m = np.array([r["obs"]["zf"].shape[0] for r in ref])
        n = np.array([r["obs"]["zn"].shape[0] for r in ref])
        res = np.array([r["resampled"] for r in ref])
        steps = {1, 2, 3, int(np.argmax(m)) + 1, len(ref)} | set(extra)
        both = np.where((m > 0) & (n > 0))[0]
        if len(both):
            steps.add(int(both[len(both) // 2]) + 1)
        steps.add(int(np.where(res)[0][0]) + 1)
        steps.add(int(np.where(~res & (m > 0))[0][5]) + 1)
        steps.add(int(np.where(res & (m >= 4))[0][-1]) + 1)
        steps.add(int(np.where(~res & (m >= 3))[0][-1]) + 1)
        return steps

    for mapname, seed in (("example_loop2", 7), ("example_loop902", 3)):
        short = mapname.replace("example_", "")
        for method, tag in (("FASTSLAM2", "fs2"), ("FASTSLAM1", "fs1")):
            a = args(mapname, method, 100, seed)
            ref, _, _, _ = run_pair(R, O, a, set(), 10 ** 9)
            steps = pick(ref, extra=(len(ref) // 2,))
            ref, snaps, preds, meta = run_pair(R, O, a, steps, 10 ** 9, pred_snaps={1, 2, 9, 10, 400})
            save_traj("traj_%s_%s_N100_s%d" % (tag, short, seed), ref, snaps, preds, meta)
    # loop902 at N = 1000: prefix deep enough that every particle carries > 40 landmarks (plain rows well populated)
    a = args("example_loop902", "FASTSLAM2", 1000, 3)
    ref, snaps, preds, meta = run_pair(R, O, a, {2, 880}, 900)
    save_traj("traj_fs2_loop902_N1000_s3", ref, snaps, preds, meta)


def make_ekf(R, name, mapname, seed, extra=()):
    """EKF-SLAM (config 1) reference run: pose, state dimension and trace(P) after every control step."""
    a = ["-m", os.path.join(ROOT, "data", mapname + ".mat"), "-method", "EKF1", "-SWITCH_SEED_RANDOM", seed] + list(extra)
    r = R.sim(a)
    xs, dims, trs, obs, true = [], [], [], [], []
    while True:
        k = r.step()
        if k < 0:
            break
        x, P = r.ekf_state()
        xs.append(x[:3].copy())
        dims.append(x.shape[0])
        trs.append(float(np.trace(P.astype(np.float64))))
        obs.append(k)
        true.append(r.true_pose()[0])
    x, P = r.ekf_state()
    r.close()
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, x=np.stack(xs).astype(f32), dim=np.array(dims, np.int32), trace=np.array(trs), observed=np.array(obs, np.int8),
                        true=np.stack(true).astype(f32), final_x=x, final_P=P)
    print(name, "control steps", len(xs), "final dim", dims[-1], "%.0f KB" % (os.path.getsize(path) / 1024))


def make_log_kats(R):
    """gaussEvaluate(v, S, logflag = 1) of the reference objects (fastslam2.cpp:154-160) on the same S, v as the
    logflag = 0 KATs: pins the log-weight extension of the oracle where the reference itself can go."""
    k = np.load(os.path.join(OUT, "kat_functions.npz"))
    d = {}
    for D in (2, 3):
        S, v = k["gauss%d_S" % D], k["gauss%d_v" % D]
        d["gauss%d_log" % D] = np.array([R.gauss_evaluate(v[i].copy(), S[i].copy(), 1) for i in range(S.shape[0])], f32)
    np.savez_compressed(os.path.join(OUT, "kat_log.npz"), **d)


def make_assoc_kats(R):
    """EKFSLAM::dataAssociate (algorithms/ekfslam.cpp:151-189) of the reference objects applied to single FastSLAM
    particles (pose known: P = blockdiag(0, Pf_j)): groups of 16 particles sharing the landmark count and the observation
    list (as the particles of one filter do), labels per particle and observation: landmark index, -1 new, -2 dropped."""
    rng = np.random.default_rng(20261004)
    d = {}
    groups = 10
    for g in range(groups):
        nf = int(rng.integers(1, 30))
        N = 16
        xv0 = (rng.normal(size=3) * [20, 20, 1.0])
        xf0 = xv0[:2] + rng.normal(size=(nf, 2)) * 25
        xv = (xv0 + rng.normal(size=(N, 3)) * [0.3, 0.3, 0.03]).astype(f32)
        xf = (xf0 + rng.normal(size=(N, nf, 2)) * 0.3).astype(f32)
        B = rng.normal(size=(N, nf, 2, 2)) * rng.uniform(0.02, 0.6, size=(N, nf, 1, 1))
        Pf = (B @ B.transpose(0, 1, 3, 2) + 1e-4 * np.eye(2)).astype(f32)
        zs = []
        for k in range(9):
            j = int(rng.integers(0, nf))
            dx, dy = xf0[j] - xv0[:2]
            scale = rng.choice([0.5, 2.0, 6.0, 30.0])
            zs.append([np.hypot(dx, dy) + rng.normal() * 0.1 * scale, np.arctan2(dy, dx) - xv0[2] + rng.normal() * 0.0175 * scale])
        for k in range(2):
            zs.append([rng.uniform(1, 60), rng.uniform(-3, 3)])
        z = np.array(zs, f32)
        lab = np.stack([R.data_associate(xv[i], xf[i], Pf[i], z, RM, 4.0, 25.0) for i in range(N)])
        d.update({"g%d_xv" % g: xv, "g%d_xf" % g: xf, "g%d_Pf" % g: Pf, "g%d_z" % g: z, "g%d_lab" % g: lab})
    d["n_groups"] = np.array(groups)
    d["gates"] = np.array([4.0, 25.0], f32)
    np.savez_compressed(os.path.join(OUT, "kat_assoc.npz"), **d)
    print("kat_assoc:", {k: int(sum((d["g%d_lab" % g] == k).sum() if k < 0 else (d["g%d_lab" % g] >= 0).sum() for g in range(groups))) for k in (0, -1, -2)})


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "assoc":
        make_assoc_kats(orc.Reference())  # per-particle association decisions (added in round 2)
    elif len(sys.argv) > 1 and sys.argv[1] == "log":
        make_log_kats(orc.Reference())  # only the logflag = 1 vectors (added in round 2)
    elif len(sys.argv) > 1 and sys.argv[1] == "maps":
        main_maps()  # example_loop2 / example_loop902 trajectories (added in round 4)
    else:
        main()
