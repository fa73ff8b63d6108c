"""Per-particle data association carried into the update (slamgpu_update_particle / slamgpu_update_labels; SURVEY.md
section 8(f4): EKFSLAM::dataAssociate, ekfslam.cpp:151-189, applied by every particle to a map of its own, which the reference's
Particle lets grow per particle, Particle.cpp:61-73).  The reference has no FastSLAM implementation of it, so the pins are:
  - particles that all agree with the known association: the step is slamgpu_update's, bit for bit (fastslam{1,2}.cpp's update);
  - particles that disagree: every particle's pose, covariance and landmarks are those of a filter in which EVERYBODY made that
    particle's decisions (the per-particle arithmetic does not know about the other particles), bit for bit, slots it did not
    open are absent, and the weight of an unexplained observation is the stated factor;
  - whole runs: the map of the best particle against the truth, beside a run with the known association."""
import numpy as np
import pytest

from conftest import load_golden as load_golden_, sim_args

pytestmark = pytest.mark.gpu
f32 = np.float32
NEW, DISCARD = -1, -2


@pytest.fixture(scope="module")
def sg():
    import slam_amd
    assert slam_amd.device_count() >= 1
    return slam_amd


def _tape(method, n, steps, seed=3):
    from slam_amd import host
    return host.make_tape(sim_args("example_webmap", method, n, seed), max_obs=steps)


def _same(a, b, what):
    for k in ("xv", "Pv", "w", "xf", "Pf"):
        x, y = np.asarray(a[k]), np.asarray(b[k])
        assert x.shape == y.shape, (what, k, x.shape, y.shape)
        assert np.array_equal(x, y, equal_nan=True), (what, k, int((x != y).sum()))


def _predicts(s, st, tape):
    for V, G, phi in np.array(st["controls"], f32).reshape(-1, 3):
        s.predict(float(V), float(G), tape["Q"], float(tape["dt"]), float(phi))


@pytest.mark.parametrize("method,name,logw", [(2, "FASTSLAM2", False), (1, "FASTSLAM1", False), (2, "FASTSLAM2", True), (1, "FASTSLAM1", True)])
@pytest.mark.parametrize("math_mode", [0, 1], ids=["strict", "fast"])
def test_unanimous_particles_take_the_known_association_step(sg, method, name, logw, math_mode):
    """Every particle carries the labels dataAssociationKnown would give (core.cpp:91-120): 60 steps of example_webmap through
    slamgpu_update_labels leave the state slamgpu_update leaves, bit for bit (poses, covariances, weights, every landmark record,
    Neff and the resampling decisions) -- on plain rows and against the compact layout the known association runs on; linear and
    log-weight contexts."""
    N, steps = 1000, 60
    tape = _tape(name, N, steps)
    kw = dict(method=method, n_effective=int(0.75 * N), rng_mode=sg.RNG_PHILOX, seed=11, math_mode=math_mode, log_weights=logw)
    a = sg.SlamGpu(N, tape["nlm"], **kw)
    b = sg.SlamGpu(N, tape["nlm"], particle_maps=True, **kw)
    for t, st in enumerate(tape["steps"]):
        zf, zn = np.array(st["zf"], f32).reshape(-1, 2), np.array(st["zn"], f32).reshape(-1, 2)
        idf = np.array(st["idf"], np.int32)
        _predicts(a, st, tape)
        _predicts(b, st, tape)
        if len(zf) + len(zn) == 0:
            continue
        a.update(zf, idf, zn, tape["R"])
        lab = np.tile(np.concatenate([idf, np.full(len(zn), NEW, np.int32)]), (N, 1))
        rep = b.update_labels(np.concatenate([zf, zn]), tape["R"], lab, p_new=1.0, census_every=4)
        assert rep["rewritten"] == len(zf) and rep["opened"] == len(zn) and rep["dropped"] == 0 and rep["dead"] == 0, (t, rep)
        a.estimate_async()
        b.estimate_async()
        if t % 20 == 19:
            _same(a.download(), b.download(), "step %d" % t)
    ea, na, ra = a.history_fetch()
    eb, nb, rb = b.history_fetch()
    da, db = a.download(), b.download()
    a.close()
    b.close()
    assert ra.any(), "the run never resampled: the genealogy was not exercised"
    assert np.array_equal(ea, eb) and np.array_equal(na, nb) and np.array_equal(ra, rb)
    _same(da, db, "end of the run")
    assert da["nf"] == db["nf"] >= 6


def _group_labels(N, idf, n_new, nf):
    """labels [N, nz] for observations [re-observed.., new..]: four kinds of particle"""
    nz = len(idf) + n_new
    lab = np.empty((N, nz), np.int32)
    base = np.concatenate([idf, np.full(n_new, NEW, np.int32)])
    for i in range(N):
        v = base.copy()
        kind = i % 4
        if kind == 1:      # ignores the first observation, does not open the new landmarks
            v[0] = DISCARD
            v[len(idf):] = DISCARD
        elif kind == 2 and len(idf) >= 2:   # matches the first observation with the SECOND observation's landmark: two claims on it
            v[0] = idf[1]
        elif kind == 3:    # calls everything it sees new (opens nothing already there) -- or, with nothing else, ignores the step
            v[:] = DISCARD
        lab[i] = v
    return lab


@pytest.mark.parametrize("method,name", [(2, "FASTSLAM2"), (1, "FASTSLAM1")])
@pytest.mark.parametrize("math_mode", [0, 1], ids=["strict", "fast"])
def test_particles_that_disagree_each_take_their_own_step(sg, method, name, math_mode):
    """After 20-odd agreed steps the particles split four ways over one step (ignore an observation; claim one landmark twice; ignore
    everything; the known association).  Every particle's pose, Pv and landmark records must be those of a filter in which ALL
    particles made its decisions -- run here through slamgpu_update with that kind's (zf, idf, zn), from the same uploaded state --
    bit for bit; slots a particle did not open hold the absent record (NaN); a particle nothing concerns keeps pose and Pv; and
    halving p_new halves a particle's weight once per observation it left unexplained."""
    N = 1024
    tape = _tape(name, N, 200)
    # the first step past the 20th with at least two re-observed landmarks and a new one: the agreed steps run up to it
    warm = next(t for t, st in enumerate(tape["steps"]) if t >= 20 and len(st["idf"]) >= 2 and len(np.array(st["zn"]).reshape(-1, 2)) >= 1)
    # (the agreed steps resample as usual; the step under test does not, so that particle i stays particle i in every run)
    kw = dict(method=method, n_effective=int(0.75 * N), rng_mode=sg.RNG_PHILOX, seed=5, math_mode=math_mode, resample=False)
    base = sg.SlamGpu(N, tape["nlm"], particle_maps=True, **dict(kw, resample=True))
    for st in tape["steps"][:warm]:
        _predicts(base, st, tape)
        zf, zn = np.array(st["zf"], f32).reshape(-1, 2), np.array(st["zn"], f32).reshape(-1, 2)
        if len(zf) + len(zn):
            base.update(zf, np.array(st["idf"], np.int32), zn, tape["R"])
    pick = tape["steps"][warm]
    state = base.download()
    nf = state["nf"]
    base.close()
    zf, zn = np.array(pick["zf"], f32).reshape(-1, 2), np.array(pick["zn"], f32).reshape(-1, 2)
    idf = np.array(pick["idf"], np.int32)
    z = np.concatenate([zf, zn])
    lab = _group_labels(N, idf, len(zn), nf)

    def run(fn, **extra):
        s = sg.SlamGpu(N, tape["nlm"], particle_maps=True, **dict(kw, **extra))
        s.upload(state)
        _predicts(s, pick, tape)
        out = fn(s)
        d = s.download()
        s.close()
        return d, out

    got, rep = run(lambda s: s.update_labels(z, tape["R"], lab, p_new=1.0, census_every=0))
    half, _ = run(lambda s: s.update_labels(z, tape["R"], lab, p_new=0.5, census_every=0))
    assert rep["rewritten"] == len(idf) and rep["opened"] == len(zn) and rep["slots"] == nf + len(zn), rep
    # what each kind of particle did, as a known-association step of everybody.  One observation per landmark and particle (the
    # first to claim it); the packet lists the landmarks in the order of the first observation that names them over ALL particles
    # (ties: by slot), and that is the order every particle meets them in
    claims, first = {}, {}
    for kind in range(4):
        c = {}
        for j, l in enumerate(lab[kind]):
            if l >= 0 and l not in c:
                c[int(l)] = j
                first[int(l)] = min(first.get(int(l), 1 << 30), j)
        claims[kind] = c
    order = sorted(first, key=lambda l: (first[l], l))
    kinds, unexplained = {}, {}
    for kind in range(4):
        ls = [l for l in order if l in claims[kind]]
        opens = [j for j in range(len(idf), len(z)) if lab[kind][j] == NEW]
        kinds[kind] = (z[[claims[kind][l] for l in ls]].reshape(-1, 2), np.array(ls, np.int32), z[opens].reshape(-1, 2))
        unexplained[kind] = len(z) - len(ls)
    assert len(kinds[0][1]) == len(idf) and len(kinds[2][1]) == len(idf) - 1 and len(kinds[3][1]) == 0 and unexplained[3] == len(z)
    for kind, (kzf, kidf, kzn) in kinds.items():
        sel = np.arange(N) % 4 == kind
        if len(kzf) + len(kzn) == 0:
            exp, _ = run(lambda s: None)
        else:
            exp, _ = run(lambda s: s.update(kzf, kidf, kzn, tape["R"]))
        assert np.array_equal(got["xv"][sel], exp["xv"][sel]), kind
        assert np.array_equal(got["Pv"][sel], exp["Pv"][sel]), kind
        # landmarks the kind holds: the expected run's; (kind 1 / 3 did not open the new slots: absent)
        nfe = exp["nf"]
        assert np.array_equal(got["xf"][sel][:, :nfe], exp["xf"][sel][:, :nfe]), kind
        assert np.array_equal(got["Pf"][sel][:, :nfe], exp["Pf"][sel][:, :nfe]), kind
        if nfe < rep["slots"]:
            assert np.isnan(got["xf"][sel][:, nfe:rep["slots"]]).all(), kind
        # weights: proportional inside the kind (each run normalises by its own total) ...
        wg, we = got["w"][sel].astype(np.float64), exp["w"][sel].astype(np.float64)
        ok = wg > 0  # (a particle that matched an observation with the wrong landmark ends at weight 0 in linear weights, both ways)
        assert np.array_equal(ok, we > 0), kind
        assert kind == 2 or ok.all(), kind
        if ok.any():
            assert np.allclose(wg[ok] / wg[ok].sum(), we[ok] / we[ok].sum(), rtol=2e-5, atol=0), kind
            # ... and the unexplained observations cost p_new each
            ratio = half["w"][sel].astype(np.float64)[ok] / wg[ok]
            ref = (half["w"][0].astype(np.float64) / got["w"][0].astype(np.float64)) * 0.5 ** (unexplained[kind] - unexplained[0])
            assert np.allclose(ratio, ref, rtol=2e-5), (kind, ratio[:3], ref)
    assert np.array_equal(half["xv"], got["xv"]) and np.array_equal(half["xf"], got["xf"], equal_nan=True)


@pytest.mark.parametrize("mapname,N,seed,math_mode,n_true", [("example_webmap", 512, 7, 1, 35), ("example_webmap", 2048, 8, 0, 35),
                                                             ("example_loop1", 512, 9, 1, 22), ("example_loop902", 1024, 7, 1, 117)])
def test_whole_runs_with_every_particle_on_its_own_association(sg, mapname, N, seed, math_mode, n_true):
    """Whole runs of the bundled maps with slamgpu_update_particle (the .ini's gates 4 / 25, the exclusion rule, a slot per
    observation 2 % of the particles call new): the best particle ends with the map's landmarks (at most two spurious ones), every
    true landmark within 1 m of one of its entries on the loop maps, and the estimate is as good as the SAME run's with the
    reference's known association (within 1.2 x + 0.05 m of its mean error).  40 + 28 such runs: profiles/particle_association_r06.txt."""
    import argparse
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("particle_assoc_probe", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools",
                                                                                      "particle_assoc_probe.py"))
    probe = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(probe)
    from slam_amd import host
    a = argparse.Namespace(method="FASTSLAM2", gate_reject=4.0, gate_augment=25.0, new_share=0.02, p_new=0.0, census=1, excl_base=2.0, excl_per_m=0.05,
                           unique_ratio=2.0, slots=4)
    r = probe.run(sg, host, mapname, N, seed, math_mode, a, False)
    k = probe.run(sg, host, mapname, N, seed, math_mode, a, True)
    assert n_true <= r["n_map"] <= n_true + 2 and r["slots"] <= n_true + 6 and r["dropped"] == 0, r
    assert r["err"] <= 1.2 * k["err"] + 0.05, (r, k)
    if mapname != "example_webmap":
        assert r["covered"] == n_true and r["stray"] == 0, r


@pytest.mark.parametrize("math_mode", [0, 1], ids=["strict", "fast"])
@pytest.mark.parametrize("name,method", [("traj_fs2_webmap_N100_s7", 2), ("traj_fs1_webmap_N100_s7", 1), ("traj_fs2_loop902_N100_s3", 2)])
def test_disagreeing_particles_against_the_oracle(sg, oracle, name, method, math_mode):
    """The same split, held to the ORACLE directly (not through slamgpu_update): from the reference's own pre-update particle sets
    (tests/golden: the snapshots of the reference objects) and its tape of normals, one slamgpu_update_labels step with four kinds
    of particle; every kind is compared with the oracle's per-particle stage (orc_update_local: fastslam{1,2}.cpp's update without
    the resampling, oracle/slam_oracle.c) run with that kind's (zf, idf, zn) -- poses, covariances, landmark records and, inside a
    kind, the weights, to the tolerances the known-association step is held to (tests/test_gpu_parity.py)."""
    from oracle import orc
    from test_gpu_parity import POSE_ATOL, close_cov, sym, compare_weights
    g = load_golden_(name)
    cap = max(40, int(g["nf"].max())) + 8
    algo = orc.Algo(method, int(g["meta_use_heading"]), int(g["meta_add_predict_noise"]), 0, int(g["meta_n_effective"]), float(g["meta_wheel_base"]),
                    float(g["meta_sigma_phi"]))
    done = 0
    for k in g["snap_steps"]:
        m, n = int(g["m"][k - 1]), int(g["n"][k - 1])
        if m < 2:
            continue
        pre = {key: g["snap%d_pre_%s" % (k, key)] for key in ("xv", "Pv", "w", "xf", "Pf")}
        N = pre["w"].shape[0]
        pre["nf"] = nf = pre["xf"].shape[1]
        zf, idf, zn = g["zf"][k - 1, :m], g["idf"][k - 1, :m].astype(np.int32), g["zn"][k - 1, :n]
        z = np.concatenate([zf, zn]).reshape(-1, 2)
        lab = _group_labels(N, idf, n, nf)
        normals = g["snap%d_normals" % k]
        s = sg.SlamGpu(N, cap, method=method, n_effective=int(g["meta_n_effective"]), use_heading=bool(g["meta_use_heading"]),
                       wheel_base=float(g["meta_wheel_base"]), sigma_phi=float(g["meta_sigma_phi"]), rng_mode=sg.RNG_TAPE, math_mode=math_mode,
                       resample=False, particle_maps=True)
        s.upload(pre)
        rep = s.update_labels(z, g["meta_R"], lab, p_new=1.0, census_every=0, normals=normals, strata=g["snap%d_sel" % k])
        got = s.download()
        s.close()
        assert rep["rewritten"] == m and rep["opened"] == n, rep
        claims, first = {}, {}
        for kind in range(4):
            c = {}
            for j, l in enumerate(lab[kind]):
                if l >= 0 and int(l) not in c:
                    c[int(l)] = j
                    first[int(l)] = min(first.get(int(l), 1 << 30), j)
            claims[kind] = c
        order = sorted(first, key=lambda l: (first[l], l))
        for kind in range(4):
            sel = np.arange(N) % 4 == kind
            ls = [l for l in order if l in claims[kind]]
            opens = [j for j in range(m, len(z)) if lab[kind][j] == NEW]
            o = orc.Particles(oracle, N, cap)
            o.set(pre)
            if ls or opens:
                o.update_local(algo, z[[claims[kind][l] for l in ls]].reshape(-1, 2), np.array(ls, np.int32), z[opens].reshape(-1, 2), g["meta_R"], normals)
            exp = o.get()
            o.close()
            tag = "%s step %d kind %d" % (name, k, kind)
            assert np.abs(got["xv"][sel] - exp["xv"][sel]).max() <= POSE_ATOL, tag
            assert close_cov(got["Pv"][sel], sym(exp["Pv"][sel])), tag
            nfe = exp["nf"]
            if nfe:
                assert np.abs(got["xf"][sel][:, :nfe] - exp["xf"][sel]).max() <= POSE_ATOL * 5, tag
                assert close_cov(got["Pf"][sel][:, :nfe], sym(exp["Pf"][sel])), tag
            if nfe < rep["slots"]:
                assert np.isnan(got["xf"][sel][:, nfe:rep["slots"]]).all(), tag
            wg, we = got["w"][sel].astype(np.float64), exp["w"][sel].astype(np.float64)
            ok = (wg > 0) & (we > 0)
            if kind != 2:
                assert ok.all(), tag
            if ok.sum() > 4:
                compare_weights((wg[ok] / wg[ok].sum()).astype(np.float32), (we[ok] / we[ok].sum()).astype(np.float32), method == 2, tag, math_mode)
        done += 1
    assert done >= 2, "no snapshot with two re-observed landmarks"


def test_a_slot_whose_holders_died_is_found_dead_and_reused(sg):
    """A landmark only a minority opened lives as long as the minority does.  Ten of 1 024 particles (weight 0, so that the next
    resampling drops them) open a slot for an observation the others discard; after the resample nobody holds it: the holders census
    (every step here) finds it dead -- out of the association, reported -- and the next landmark everybody opens REUSES the slot
    instead of growing the map; every particle then holds it."""
    N = 1024
    tape = _tape("FASTSLAM2", N, 40)
    kw = dict(method=2, n_effective=N, rng_mode=sg.RNG_PHILOX, seed=3, math_mode=1, particle_maps=True)   # NEFFECTIVE = N: every step resamples
    s = sg.SlamGpu(N, 64, **kw)
    for st in tape["steps"][:20]:
        _predicts(s, st, tape)
        zf, zn = np.array(st["zf"], f32).reshape(-1, 2), np.array(st["zn"], f32).reshape(-1, 2)
        if len(zf) + len(zn):
            s.update(zf, np.array(st["idf"], np.int32), zn, tape["R"])
    d = s.download()
    nf = d["nf"]
    minority = np.arange(10)
    d["w"] = d["w"].copy()
    d["w"][minority] = 0.0
    d["w"] /= d["w"].sum()
    s.upload(d)
    # one observation far from everything: the minority calls it new, everybody else discards it
    z = np.array([[25.0, 0.3]], f32)
    lab = np.full((N, 1), DISCARD, np.int32)
    lab[minority, 0] = NEW
    r1 = s.update_labels(z, tape["R"], lab, new_share=0.0, p_new=1.0, census_every=1)
    assert r1["opened"] == 1 and r1["slots"] == nf + 1 and r1["dead"] == 0, r1
    after = s.download()   # (materialises: the resample has happened, the weightless minority is gone)
    assert after["nf"] == nf + 1 and np.isnan(after["xf"][:, nf, 0]).all(), "somebody still holds the minority's landmark"
    # a step that opens nothing: the census sees the slot without holders
    lab2 = np.full((N, 1), DISCARD, np.int32)
    r2 = s.update_labels(z, tape["R"], lab2, new_share=0.0, p_new=1.0, census_every=1)
    assert r2["census"] == 1 and r2["dead"] == 1 and r2["opened"] == 0 and r2["slots"] == nf + 1, r2
    # everybody opens a landmark: the dead slot is taken, the map does not grow
    lab3 = np.full((N, 1), NEW, np.int32)
    r3 = s.update_labels(z, tape["R"], lab3, new_share=0.5, p_new=1.0, census_every=1)
    assert r3["opened"] == 1 and r3["reused"] == 1 and r3["dead"] == 0 and r3["slots"] == nf + 1, r3
    end = s.download()
    s.close()
    assert end["nf"] == nf + 1 and not np.isnan(end["xf"][:, nf, 0]).any()
    assert np.isfinite(end["xv"]).all() and abs(end["w"].sum(dtype=np.float64) - 1.0) < 1e-4


def test_capacity_layout_and_argument_errors_are_return_codes(sg):
    """The per-particle entry points on the edges: a context of 40..256 landmarks created WITHOUT the flag is moved to plain rows at
    the first call and works; a small compact context is refused with the flag's name; an observation that needs a slot when none is
    left is dropped and reported, not an error; bad options are SLAMGPU_ERR_INVALID; an exhaustive exclusion scan that would run for
    seconds is SLAMGPU_ERR_CAPACITY."""
    N = 512
    tape = _tape("FASTSLAM2", N, 25)
    R = tape["R"]
    kw = dict(method=2, n_effective=int(0.75 * N), rng_mode=sg.RNG_PHILOX, seed=2, math_mode=1)
    # (i) mid-size capacity without the flag: demoted on the fly
    s = sg.SlamGpu(N, 100, **kw)
    for st in tape["steps"]:
        _predicts(s, st, tape)
        z = np.concatenate([np.array(st["zf"], f32).reshape(-1, 2), np.array(st["zn"], f32).reshape(-1, 2)])
        if len(z):
            s.update_particle(z, R, 4.0, 25.0, new_share=0.02, p_new=10.0, excl=(2.0, 0.05, 2.0))
    d = s.download()
    assert 3 <= d["nf"] <= 12 and np.isfinite(d["xv"]).all()
    # (ii) bad options
    z1 = np.array([[10.0, 0.1]], f32)
    for bad in (dict(p_new=0.0), dict(new_share=1.5), dict(census_every=-1), dict(excl=(-1.0, 0.0, 2.0))):
        with pytest.raises(sg.SlamGpuError) as e:
            s.update_particle(z1, R, 4.0, 25.0, **dict(dict(p_new=1.0), **bad))
        assert e.value.code == -1, (bad, e.value)
    with pytest.raises(sg.SlamGpuError) as e:      # the exclusion rule scans exhaustively: not through the grid
        s.update_particle(z1, R, 4.0, 25.0, mode=sg.capi.ASSOC_GRID, p_new=1.0, excl=(2.0, 0.05, 2.0))
    assert e.value.code == -1 and "exclusion" in str(e.value)
    s.close()
    # (iii) a compact context (capacity below 40) is refused, by name
    s = sg.SlamGpu(N, 35, **kw)
    with pytest.raises(sg.SlamGpuError) as e:
        s.update_particle(z1, R, 4.0, 25.0, p_new=1.0)
    assert e.value.code == -1 and "SLAMGPU_FLAG_PARTICLE_MAPS" in str(e.value)
    s.close()
    # (iv) no slot left: the observation is dropped and said so
    s = sg.SlamGpu(N, 40, particle_maps=True, **kw)
    lab = np.full((N, 50), NEW, np.int32)
    z50 = np.stack([np.linspace(5, 50, 50), np.linspace(-1, 1, 50)], axis=1).astype(f32)
    r = s.update_labels(z50, R, lab, new_share=0.5, p_new=1.0)
    assert r["opened"] == 40 and r["dropped"] == 10 and r["slots"] == 40, r
    r = s.update_labels(z50[:3], R, lab[:, :3], new_share=0.5, p_new=1.0)
    assert r["opened"] == 0 and r["dropped"] == 3 and r["slots"] == 40, r
    d = s.download()
    assert d["nf"] == 40 and np.isfinite(d["xf"]).all()
    s.close()
    # (v) an exhaustive exclusion scan of 4 096 x 4 000 x 4 000 gates is refused before it is launched
    s = sg.SlamGpu(4096, 4000, particle_maps=True, **dict(kw, n_effective=3072))
    lab = np.full((4096, 4000), NEW, np.int32)
    zbig = np.stack([np.linspace(5, 60, 4000), np.linspace(-3, 3, 4000)], axis=1).astype(f32)
    s.update_labels(zbig, R, lab, new_share=0.5, p_new=1.0)
    with pytest.raises(sg.SlamGpuError) as e:
        s.update_particle(zbig, R, 4.0, 25.0, p_new=1.0, excl=(2.0, 0.05, 2.0))
    assert e.value.code == -3 and "exclusion" in str(e.value), e.value
    s.close()
