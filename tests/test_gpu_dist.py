"""Distributed contexts (slamgpu_dist_*: peer-mapped state, one launch + one all-gather per step, nothing migrates) on
one GPU: G logical shards must reproduce the single-context run -- same Neff, decision, history and bit-identical
particle state -- for any G, both methods, both builds, small (compact genealogy) and large (plain rows) maps."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import sim_args

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_single(sg, tp, Np, method, math_mode, seed, check_at=()):
    s = sg.SlamGpu(Np, tp["nlm"], method=method, n_effective=int(0.75 * Np), rng_mode=sg.RNG_PHILOX, seed=seed, math_mode=math_mode)
    Q, R, dt = tp["Q"], tp["R"], float(tp["dt"])
    mid = {}
    for k, st in enumerate(tp["steps"]):
        s.step(np.array(st["controls"], np.float32).reshape(-1, 3), Q, dt, st["zf"], st["idf"], st["zn"], R)
        if k in check_at:
            mid[k] = s.download()
    hist = s.history_fetch()
    d = s.download()
    s.close()
    return d, hist, mid


def run_dist(sg, tp, Np, G, method, math_mode, seed, check_at=(), push=False):
    from slam_amd.dist import DistFilter
    f = DistFilter.local(G, Np // G, tp["nlm"], method=method, n_effective=int(0.75 * Np), seed=seed, math_mode=math_mode)
    if push:
        assert f.use_push(fold=(push == "fold"))
    assert sum(c.dist_remote_reads() for c in f.ctx) == 0   # (the counter is kept from the first time it is asked for)
    Q, R, dt = tp["Q"], tp["R"], float(tp["dt"])
    mid = {}
    cat = lambda parts: {k: np.concatenate([p[k] for p in parts]) for k in ("xv", "Pv", "w", "xf", "Pf")}
    for k, st in enumerate(tp["steps"]):
        f.step(np.array(st["controls"], np.float32).reshape(-1, 3), Q, dt, st["zf"], st["idf"], st["zn"], R)
        if k in check_at:
            mid[k] = cat(f.download())
    hist = f.history_fetch()
    d = cat(f.download())
    if push:
        assert f.collective_ok()
    # ancestors read out of another shard's memory (slamgpu_dist_remote_reads): none with one shard; with several, stratified
    # ancestors sit next to their offspring, so only the particles around the shard boundaries cross
    remote = sum(c.dist_remote_reads() for c in f.ctx)
    resampled = int(np.count_nonzero(hist[2])) * Np
    if G == 1:
        assert remote == 0
    elif resampled:
        assert 0 < remote <= 0.25 * resampled, (G, remote, resampled)
    f.close()
    return d, hist, mid


def same_state(a, b, where):
    for key in ("xv", "Pv", "w", "xf", "Pf"):
        assert np.array_equal(a[key].view(np.uint32), b[key].view(np.uint32)), (where, key)


def same_history(ha, hb):
    xa, na, ra = ha
    xb, nb, rb = hb
    assert len(xa) == len(xb) > 0
    assert np.array_equal(na, nb) and np.array_equal(ra, rb)
    assert np.allclose(xa, xb, rtol=0, atol=1e-12)


@pytest.mark.parametrize("G", [1, 2, 4, 8])
@pytest.mark.parametrize("math_mode", [0, 1])
def test_logical_shards_match_single_context_webmap(G, math_mode):
    import slam_amd as sg
    from slam_amd import host
    Np, nobs = 4096, 70
    tp = host.make_tape(sim_args("example_webmap", "FASTSLAM2", Np, 3), max_obs=nobs)
    ref, href, mref = run_single(sg, tp, Np, sg.FASTSLAM2, math_mode, 9, check_at=(20, 21))
    got, hgot, mgot = run_dist(sg, tp, Np, G, sg.FASTSLAM2, math_mode, 9, check_at=(20, 21))
    assert href[2].sum() >= 3  # several resamples: ancestors cross shard boundaries
    same_history(href, hgot)
    for k in mref:  # reading the set in the middle of a run (a settle launch) must not change what follows
        same_state(mref[k], mgot[k], k)
    same_state(ref, got, "final")


@pytest.mark.parametrize("G", [2, 4])
def test_logical_shards_match_single_context_fastslam1(G):
    import slam_amd as sg
    from slam_amd import host
    Np, nobs = 2048, 50
    tp = host.make_tape(sim_args("example_webmap", "FASTSLAM1", Np, 5), max_obs=nobs)
    ref, href, _ = run_single(sg, tp, Np, sg.FASTSLAM1, 1, 11)
    got, hgot, _ = run_dist(sg, tp, Np, G, sg.FASTSLAM1, 1, 11)
    assert href[2].sum() >= 1
    same_history(href, hgot)
    same_state(ref, got, "final")


@pytest.mark.parametrize("G", [2, 4])
def test_logical_shards_match_single_context_many_landmarks(G, tmp_path):
    """a map beyond the compact genealogy (plain rows, device packets, copy roles run by helper blocks)"""
    import slam_amd as sg
    from conftest import DATA
    from slam_amd import host
    Np, nobs = 2048, 60
    lm = host.synthetic_landmarks(12345, 1000, -130, 100, -100, 90)
    h = host.HostSim(sim_args("example_webmap", "FASTSLAM2", 100, 7))
    _, wp = h.map()
    h.close()
    mp = str(tmp_path / "syn1000.mat")
    host.write_map(mp, lm, wp)
    ini = open(os.path.join(DATA, "example_webmap.ini")).read().replace("MAX_RANGE           = 60.0", "MAX_RANGE           = 20.0")
    open(str(tmp_path / "syn1000.ini"), "w").write(ini)
    tp = host.make_tape(["-m", mp, "-method", "FASTSLAM2", "-NPARTICLES", Np, "-NEFFECTIVE", int(0.75 * Np), "-SWITCH_SEED_RANDOM", 4],
                        max_obs=nobs)
    assert tp["nlm"] > 40
    ref, href, _ = run_single(sg, tp, Np, sg.FASTSLAM2, 1, 13)
    got, hgot, _ = run_dist(sg, tp, Np, G, sg.FASTSLAM2, 1, 13)
    assert href[2].sum() >= 2
    same_history(href, hgot)
    same_state(ref, got, "final")


def test_reading_an_unsettled_distributed_context_fails_loudly():
    import slam_amd as sg
    from slam_amd import host
    from slam_amd.dist import DistFilter
    Np = 1024
    tp = host.make_tape(sim_args("example_webmap", "FASTSLAM2", Np, 3), max_obs=3)
    f = DistFilter.local(2, Np // 2, tp["nlm"], method=sg.FASTSLAM2, seed=1)
    st = tp["steps"][0]
    f.step(np.array(st["controls"], np.float32).reshape(-1, 3), tp["Q"], float(tp["dt"]), st["zf"], st["idf"], st["zn"], tp["R"])
    with pytest.raises(sg.SlamGpuError, match="slamgpu_dist_settle"):
        f.ctx[0].download()
    # ... and the single-context entry points refuse a distributed context (they would skip the all-gather)
    with pytest.raises(sg.SlamGpuError, match="slamgpu_dist_step"):
        f.ctx[0].update(st["zf"], st["idf"], st["zn"], tp["R"])
    with pytest.raises(sg.SlamGpuError, match="slamgpu_dist_step"):
        f.ctx[0].shard_update(st["zf"], st["idf"], st["zn"], tp["R"])
    f.close()


def test_two_processes_share_one_gpu_over_hip_ipc():
    """one process per shard (the production launch shape), both on this GPU: state arrays mapped through hipIpc, the
    all-gather through torch.distributed (gloo on CPU copies of the totals: RCCL refuses two ranks on one device)"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29653")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dist_ipc_check.py")], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "DIST_IPC_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.parametrize("G", [1, 2, 4])
def test_group_api_logical_shards_match_single_context(G):
    """slamgpu_dist_group_* (what slam-backend -gpus k drives): k logical shards on this GPU, one stream, the copy-kernel
    all-gather; no host wait anywhere in the step loop"""
    import slam_amd as sg
    from slam_amd import host
    Np, nobs = 4096, 70
    tp = host.make_tape(sim_args("example_webmap", "FASTSLAM2", Np, 3), max_obs=nobs)
    ref, href, _ = run_single(sg, tp, Np, sg.FASTSLAM2, 1, 9)
    g = sg.DistGroup(G, Np // G, tp["nlm"], method=sg.FASTSLAM2, n_effective=int(0.75 * Np), seed=9, math_mode=1)
    for st in tp["steps"]:
        g.step(np.array(st["controls"], np.float32).reshape(-1, 3), tp["Q"], float(tp["dt"]), st["zf"], st["idf"], st["zn"], tp["R"])
    hgot = g.history_fetch()
    got = g.download()
    g.close()
    same_history(href, hgot)
    same_state(ref, got, "final")


def test_rccl_communicator_inside_the_library_one_rank():
    """slamgpu_dist_comm_init + the all-gather enqueued by slamgpu_dist_step itself (RCCL bound with dlopen), world size 1
    (one GPU here); the 8-rank run is the driver's scaling bench"""
    import slam_amd as sg
    from slam_amd import host
    Np, nobs = 2048, 40
    tp = host.make_tape(sim_args("example_webmap", "FASTSLAM2", Np, 3), max_obs=nobs)
    ref, href, _ = run_single(sg, tp, Np, sg.FASTSLAM2, 1, 9)
    c = sg.SlamGpu(Np, tp["nlm"], method=sg.FASTSLAM2, n_effective=int(0.75 * Np), seed=9, math_mode=1, rng_mode=sg.RNG_PHILOX)
    c.dist_connect(1, 0, [c.dist_export()])
    c.dist_comm_init(sg.dist_comm_id(), 1, 0)
    for st in tp["steps"]:
        c.dist_step(np.array(st["controls"], np.float32).reshape(-1, 3), tp["Q"], float(tp["dt"]), st["zf"], st["idf"], st["zn"], tp["R"])
    c.dist_settle()
    raw, neff, res, _ = c.shard_estimate_fetch_full()
    got = c.download()
    c.close()
    assert np.array_equal(neff, href[1]) and np.array_equal(res.astype(bool), href[2])
    assert np.allclose(raw[:, :2] / Np, href[0][:, :2], rtol=0, atol=1e-12)
    same_state(ref, got, "final")


def test_logical_shards_beyond_two_totals_per_thread():
    """more than 512 blocks in the gathered table: the scan walks several totals per thread across shard boundaries"""
    import slam_amd as sg
    from slam_amd import host
    Np, nobs = 536 * 256, 40
    tp = host.make_tape(sim_args("example_webmap", "FASTSLAM2", Np, 3), max_obs=nobs)
    ref, href, _ = run_single(sg, tp, Np, sg.FASTSLAM2, 1, 9)
    got, hgot, _ = run_dist(sg, tp, Np, 4, sg.FASTSLAM2, 1, 9)
    assert href[2].sum() >= 3
    same_history(href, hgot)
    same_state(ref, got, "final")


@pytest.mark.parametrize("n", [100096, 125184])
def test_logical_shards_at_the_scaling_bench_size(n):
    """the driver's N=8 scaling run in miniature time but at full width: 8 shards x 100 096 particles (3 128 gathered block
    totals scanned in LDS by every block, the 25 KB prefix beside the staged records) against one context of 800 768; and
    BASELINE config 4's shape, 8 x 125 184 = 1 001 472 particles"""
    import slam_amd as sg
    from slam_amd import host
    G, nobs = 8, 45
    Np = G * n
    tp = host.make_tape(sim_args("example_webmap", "FASTSLAM2", Np, 7), max_obs=nobs)
    ref, href, _ = run_single(sg, tp, Np, sg.FASTSLAM2, 1, 7)
    g = sg.DistGroup(G, n, tp["nlm"], method=sg.FASTSLAM2, n_effective=int(0.75 * Np), seed=7, math_mode=1)
    for st in tp["steps"]:
        g.step(np.array(st["controls"], np.float32).reshape(-1, 3), tp["Q"], float(tp["dt"]), st["zf"], st["idf"], st["zn"], tp["R"])
    hgot = g.history_fetch()
    got = g.download()
    g.close()
    assert href[2].sum() >= 3
    same_history(href, hgot)
    same_state(ref, got, "final")


@pytest.mark.parametrize("mode", [True, "fold"])
@pytest.mark.parametrize("G", [1, 2])
def test_push_collective_matches_single_context(G, mode):
    """SLAMGPU_DIST_PUSH: block totals stored straight into every shard's table by the update launch, a one-wave flag kernel
    as the barrier (each logical shard on a stream of its own); mid-run reads included.  Two shards at most inside one
    process on one device: the runtime maps the streams of a process onto a few hardware queues, and a shard whose flag
    store queues behind the wave that waits for it never arrives (the bounded spin reports it: use_push() says no).  One
    process per GPU -- the deployment -- has no such sharing: see the two-process test below."""
    import slam_amd as sg
    from slam_amd import host
    Np, nobs = 4096, 70
    tp = host.make_tape(sim_args("example_webmap", "FASTSLAM2", Np, 3), max_obs=nobs)
    ref, href, mref = run_single(sg, tp, Np, sg.FASTSLAM2, 1, 9, check_at=(20,))
    got, hgot, mgot = run_dist(sg, tp, Np, G, sg.FASTSLAM2, 1, 9, check_at=(20,), push=mode)
    assert href[2].sum() >= 3
    same_history(href, hgot)
    same_state(mref[20], mgot[20], 20)
    same_state(ref, got, "final")


@pytest.mark.parametrize("mode", ["push", "fold"])
def test_push_collective_two_processes_over_hip_ipc(mode):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29657", SLAM_DIST_PUSH=mode)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dist_ipc_check.py")], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "DIST_IPC_OK " + mode in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_flag_barrier_reports_a_peer_that_never_arrives():
    """the barrier's spin is bounded: a shard whose peer never stores its flag gets an error word, not a hang, and later
    barriers of that context do not wait again"""
    import time
    import slam_amd as sg
    from slam_amd import host
    from slam_amd.dist import DistFilter
    Np = 1024
    tp = host.make_tape(sim_args("example_webmap", "FASTSLAM2", Np, 3), max_obs=2)
    f = DistFilter.local(2, Np // 2, tp["nlm"], method=sg.FASTSLAM2, seed=1)
    t0 = time.time()
    _, ok = f.ctx[0].dist_handshake_test(3)  # shard 1 never takes part
    assert not ok and time.time() - t0 < 20.0
    assert not f.ctx[0].dist_collective_ok()
    f.close()


@pytest.mark.parametrize("mode", [False, True, "fold"])
def test_random_reads_between_distributed_steps_do_not_change_results(mode):
    """observer calls (history fetch, download, settle) at random points of a distributed run -- under each collective --
    leave every later result identical to the single-context run"""
    import slam_amd as sg
    from slam_amd import host
    from slam_amd.dist import DistFilter
    Np, nobs, G = 4096, 90, 2
    tp = host.make_tape(sim_args("example_webmap", "FASTSLAM2", Np, 3), max_obs=nobs)
    ref, href, _ = run_single(sg, tp, Np, sg.FASTSLAM2, 1, 5)
    f = DistFilter.local(G, Np // G, tp["nlm"], method=sg.FASTSLAM2, n_effective=int(0.75 * Np), seed=5, math_mode=1)
    if mode:
        assert f.use_push(fold=(mode == "fold"))
    rng = np.random.default_rng(11)
    hs = []
    for st in tp["steps"]:
        f.step(np.array(st["controls"], np.float32).reshape(-1, 3), tp["Q"], float(tp["dt"]), st["zf"], st["idf"], st["zn"], tp["R"])
        r = rng.integers(0, 12)
        if r == 0:
            hs.append(f.history_fetch())
        elif r == 1:
            f.download()
        elif r == 2:
            f.settle()
            f.settle()  # (idempotent)
        elif r == 3:
            assert f.nf() == st["nf_before"] + st["zn"].shape[0]
    hs.append(f.history_fetch())
    got = {k: np.concatenate([p[k] for p in f.download()]) for k in ("xv", "Pv", "w", "xf", "Pf")}
    if mode:
        assert f.collective_ok()
    f.close()
    hgot = [np.concatenate([h[j] for h in hs]) for j in range(3)]
    same_history(href, hgot)
    same_state(ref, got, "final")
