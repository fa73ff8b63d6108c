"""Sharded path on one GPU: G logical shards (G slamgpu contexts + LocalComm) must reproduce the single-context
result exactly — same Neff, same decision, same ancestors, bit-identical particle state — for any G."""
import numpy as np
import pytest

from conftest import sim_args

pytestmark = pytest.mark.gpu
N = 2048
NOBS = 24


@pytest.fixture(scope="module")
def tape():
    from slam_amd import host
    return host.make_tape(sim_args("example_webmap", "FASTSLAM2", N, 7), max_obs=NOBS)


def run_single(sg, tape, rng_mode, tapes=None, **kw):
    s = sg.SlamGpu(N, tape["nlm"], method=sg.FASTSLAM2, n_effective=int(0.75 * N), rng_mode=rng_mode, seed=7, **kw)
    out = []
    for k, st in enumerate(tape["steps"]):
        for (V, G, phi) in st["controls"]:
            s.predict(V, G, tape["Q"], float(tape["dt"]), phi)
        if tapes:
            s.update(st["zf"], st["idf"], st["zn"], tape["R"], tapes[k][0], tapes[k][1])
        else:
            s.update(st["zf"], st["idf"], st["zn"], tape["R"])
        ne, did, ws = s.stats()
        d = s.download()
        out.append(dict(neff=ne, res=did, wsum=ws, est=s.estimate(), keep=s.ancestors(), **d))
    s.close()
    return out


def run_sharded(sg, tape, G, rng_mode, tapes=None):
    from slam_amd.sharded import GpuEngine, LocalComm, ShardedFilter
    n = N // G
    eng = [GpuEngine(g, G, n, tape["nlm"], method=sg.FASTSLAM2, n_effective=int(0.75 * N), rng_mode=rng_mode, seed=7) for g in range(G)]
    flt = ShardedFilter(eng, LocalComm(eng), G)
    out = []
    for k, st in enumerate(tape["steps"]):
        for (V, Gs, phi) in st["controls"]:
            flt.predict(V, Gs, tape["Q"], float(tape["dt"]), phi)
        if tapes:
            normals = [tapes[k][0][g * n:(g + 1) * n] for g in range(G)]
            plan = flt.update(st["zf"], st["idf"], st["zn"], tape["R"], normals, tapes[k][1])
        else:
            plan = flt.update(st["zf"], st["idf"], st["zn"], tape["R"])
        ds = [e.ctx.download() for e in eng]
        cat = lambda key: np.concatenate([d[key] for d in ds])
        keep = np.concatenate([e.ctx.ancestors() if plan.resampled else np.arange(g * n, (g + 1) * n, dtype=np.int32)
                               for g, e in enumerate(eng)])
        out.append(dict(neff=np.float32(plan.neff), res=bool(plan.resampled), wsum=plan.wsum, est=flt.estimate(), keep=keep,
                        xv=cat("xv"), Pv=cat("Pv"), w=cat("w"), xf=cat("xf"), Pf=cat("Pf")))
    moved = flt.exchanged_records
    flt.close()
    return out, moved


@pytest.mark.parametrize("G", [2, 4, 8])
def test_logical_shards_match_single_context_philox(tape, G):
    import slam_amd as sg
    ref = run_single(sg, tape, sg.RNG_PHILOX)
    got, moved = run_sharded(sg, tape, G, sg.RNG_PHILOX)
    assert any(r["res"] for r in ref) and moved > 0
    for k, (a, b) in enumerate(zip(ref, got)):
        assert a["res"] == b["res"] and a["neff"] == b["neff"] and a["wsum"] == b["wsum"], k
        assert np.array_equal(a["keep"], b["keep"]), k
        for key in ("xv", "Pv", "w", "xf", "Pf"):
            assert np.array_equal(a[key].view(np.uint32), b[key].view(np.uint32)), (k, key)
        assert np.allclose(a["est"], b["est"], rtol=0, atol=1e-12), k


def test_logical_shards_tape_draws_and_the_parity_context(tape):
    """The caller's draws (TAPE) through the sharded path, and the ONE rule about shard counts: every run
    through the shard / distributed entry points scans the block totals in double, whatever the number of shards -- G = 1
    included -- so its results do not depend on G; a SINGLE context in the parity configuration (strict build, the caller's
    draws, at most 5 000 particles, driven through slamgpu_update) replays the reference's float32 order of operations in its
    resampling stage instead (core.cpp:718-824: that is what makes its ancestors the reference's bits), and
    SLAMGPU_FLAG_NO_REFERENCE_RESAMPLE gives it the shards' arithmetic.  All three statements are checked here."""
    import slam_amd as sg
    rng = np.random.default_rng(5)
    tapes = []
    for st in tape["steps"]:
        normals = rng.normal(size=(N, 3)).astype(np.float32)
        sel = ((np.arange(N) + rng.uniform(size=N)) / N).astype(np.float32)
        tapes.append((normals, sel))
    one, _ = run_sharded(sg, tape, 1, sg.RNG_TAPE, tapes)
    got, _ = run_sharded(sg, tape, 4, sg.RNG_TAPE, tapes)
    same = run_single(sg, tape, sg.RNG_TAPE, tapes, reference_resample=False)
    assert any(r["res"] for r in one)
    for ref in (one, same):   # 1 shard == 4 shards == a single context with the shards' arithmetic, bit for bit
        for k, (a, b) in enumerate(zip(ref, got)):
            assert a["res"] == b["res"] and a["neff"] == b["neff"], k
            assert np.array_equal(a["keep"], b["keep"]), k
            for key in ("xv", "w", "xf"):
                assert np.array_equal(a[key].view(np.uint32), b[key].view(np.uint32)), (k, key)
    # the parity context (default): float32 sums in the reference's order: Neff agrees to float32 rounding, the decision is the
    # same, and a handful of strata next to a cumulative-sum boundary pick the neighbour -- up to the first such step the two
    # runs are the same run
    par = run_single(sg, tape, sg.RNG_TAPE, tapes)
    seen = 0
    for k, (a, b) in enumerate(zip(par, got)):
        assert a["res"] == b["res"], k
        assert abs(float(a["neff"]) / float(b["neff"]) - 1.0) <= 2e-6, (k, a["neff"], b["neff"])
        if a["res"]:
            d = np.abs(a["keep"].astype(np.int64) - b["keep"])
            assert d.max() <= 1 and np.count_nonzero(d) <= 8, (k, np.count_nonzero(d))
            seen += 1
            if d.any():
                break
        for key in ("xv", "xf"):
            assert np.array_equal(a[key].view(np.uint32), b[key].view(np.uint32)), (k, key)
    assert seen >= 1


def test_torch_runtime_coexists():
    """bench.py --gpus N imports torch (its bundled HIP runtime) BEFORE libslamgpu so that both share one runtime;
    checked in a fresh process (this one has already initialised /opt/rocm's runtime through libslamgpu)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "coexist_check.py"), "torch_first"], cwd=root,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "COEXIST_OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("G,pool_cap", [(1, None), (2, None), (4, None), (4, "24"), (8, "3")])
def test_sharded_steps_without_downloads_match_single_context(G, pool_cap, monkeypatch):
    """The production loop: ShardedFilter.step() per observation step, nothing read back in between, so the local
    offspring of every resample stay a lazy gather until the next update launch and the boundary offspring travel
    through pack / all-to-all / unpack into the arrival pool (or, when the pool is full, a settling unpack).  Final state
    and the whole estimate history must equal the single-context run."""
    import slam_amd as sg
    from slam_amd import host
    from slam_amd.sharded import GpuEngine, LocalComm, ShardedFilter
    if pool_cap:  # a tiny arrival pool: most steps with arrivals must fall back to settling the whole shard
        monkeypatch.setenv("SLAMGPU_POOL_CAP", pool_cap)
    Np, nobs = 4096, 70
    tp = host.make_tape(sim_args("example_webmap", "FASTSLAM2", Np, 3), max_obs=nobs)
    Q, R, dt = tp["Q"], tp["R"], float(tp["dt"])
    ctl = [np.array(st["controls"], np.float32).reshape(-1, 3) for st in tp["steps"]]

    s = sg.SlamGpu(Np, tp["nlm"], method=sg.FASTSLAM2, n_effective=int(0.75 * Np), rng_mode=sg.RNG_PHILOX, seed=9, math_mode=1)
    for k, st in enumerate(tp["steps"]):
        s.step(ctl[k], Q, dt, st["zf"], st["idf"], st["zn"], R)
    est_ref, _, res_ref = s.history_fetch()
    ref = s.download()
    s.close()
    assert 5 < res_ref.sum() < nobs

    n = Np // G
    eng = [GpuEngine(g, G, n, tp["nlm"], method=sg.FASTSLAM2, n_effective=int(0.75 * Np), rng_mode=sg.RNG_PHILOX, seed=9, math_mode=1)
           for g in range(G)]
    flt = ShardedFilter(eng, LocalComm(eng), G)
    res = []
    for k, st in enumerate(tp["steps"]):
        plan = flt.step(ctl[k], Q, dt, st["zf"], st["idf"], st["zn"], R)
        res.append(bool(plan.resampled))
    est = flt.estimate_fetch()
    ds = [e.ctx.download() for e in eng]
    moved = flt.exchanged_records
    flt.close()
    assert res == [bool(r) for r in res_ref]
    assert G == 1 or moved > 0
    assert np.allclose(est, est_ref, rtol=0, atol=1e-9)
    for key in ("xv", "Pv", "w", "xf", "Pf"):
        got = np.concatenate([d[key] for d in ds])
        assert np.array_equal(got.view(np.uint32), ref[key].view(np.uint32)), key


def test_config4_size_eight_shards_match_one_context():
    """BASELINE configs[3] at its real size: FASTSLAM2, 8 shards x 125 184 particles (1 001 472 ~ 10^6; shards are
    multiples of 256) as logical shards on one GPU (LocalComm: device-to-device copies stand in for the collectives)
    against ONE context holding all of them (large-context path: scan_kernel): resample decisions, pose estimates
    and the final particle state bit-identical, headline (fast) build, Philox noise."""
    import os
    import slam_amd as sg
    from slam_amd import host
    from slam_amd.sharded import GpuEngine, LocalComm, ShardedFilter
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    G, n, nobs = 8, 125184, 60
    Ntot = n * G
    tape = host.make_tape(["-m", os.path.join(root, "data", "example_webmap.mat"), "-method", "FASTSLAM2", "-NPARTICLES", Ntot,
                           "-NEFFECTIVE", int(0.75 * Ntot), "-SWITCH_SEED_RANDOM", 7], max_obs=nobs)
    Q, R, dt = tape["Q"], tape["R"], float(tape["dt"])
    ctl = [np.array(st["controls"], np.float32).reshape(-1, 3) for st in tape["steps"]]
    s = sg.SlamGpu(Ntot, tape["nlm"], method=2, n_effective=int(0.75 * Ntot), rng_mode=sg.RNG_PHILOX, seed=7, math_mode=1)
    for k, st in enumerate(tape["steps"]):
        s.step(ctl[k], Q, dt, st["zf"], st["idf"], st["zn"], R)
    e1, _, r1 = s.history_fetch()
    ref = s.download()
    s.close()
    eng = [GpuEngine(g, G, n, tape["nlm"], method=2, n_effective=int(0.75 * Ntot), rng_mode=sg.RNG_PHILOX, seed=7, math_mode=1)
           for g in range(G)]
    flt = ShardedFilter(eng, LocalComm(eng), G)
    res = [bool(flt.step(ctl[k], Q, dt, st["zf"], st["idf"], st["zn"], R).resampled) for k, st in enumerate(tape["steps"])]
    est = flt.estimate_fetch()
    ds = [e.ctx.download() for e in eng]
    moved = flt.exchanged_records
    flt.close()
    assert res == [bool(x) for x in r1] and 5 < sum(res) < nobs
    assert moved > 0  # offspring really crossed shard boundaries
    assert np.abs(est - e1).max() < 1e-9
    for key in ("xv", "Pv", "w", "xf", "Pf"):
        got = np.concatenate([d[key] for d in ds])
        assert np.array_equal(got.view(np.uint32), ref[key].view(np.uint32)), key
