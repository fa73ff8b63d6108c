"""The drop-in binary on the GPU: slam_amd/bin/slam-backend -method FASTSLAM{1,2} (the reference's command line,
SLAMBackendApplication.cpp:40-89) restates FastSLAM{1,2}Wrapper::run (fastslam2wrapper.cpp:31-122,
fastslam1wrapper.cpp:32-113) in C++ over the slamgpu C ABI.  With -rng parity it feeds the libc rand() tape in the
reference's draw order, so its logged per-step estimates (ParticleSLAMWrapper::computeEstimatedPosition) must follow the
reference's golden trajectory: to 1 mm until the first resample whose ancestors may differ at a cumulative-sum boundary,
statistically (error against the true path no worse than the reference's) afterwards."""
import os
import subprocess

import numpy as np
import pytest

from conftest import DATA, load_golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(DATA)
EXE = os.path.join(ROOT, "slam_amd", "bin", "slam-backend")


def run_backend(tmp_path, method, math, extra=(), maxsteps=4000, mapname="example_webmap", seed=7):
    log = str(tmp_path / ("%s_%s.csv" % (method, math)))
    cmd = [EXE, "-m", os.path.join(DATA, mapname + ".mat"), "-method", method, "-rng", "parity", "-math", math,
           "-NPARTICLES", "100", "-NEFFECTIVE", "75", "-SWITCH_SEED_RANDOM", str(seed), "-log", log, "-maxsteps", str(maxsteps), *extra]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-800:] + r.stderr[-800:]
    rows = np.loadtxt(log, delimiter=",", skiprows=1)
    return r.stdout, rows


@pytest.mark.parametrize("math", ["strict", "fast"])
@pytest.mark.parametrize("method,golden,mapname,seed", [("FASTSLAM2", "traj_fs2_webmap_N100_s7", "example_webmap", 7),
                                                        ("FASTSLAM1", "traj_fs1_webmap_N100_s7", "example_webmap", 7),
                                                        ("FASTSLAM2", "traj_fs2_loop2_N100_s7", "example_loop2", 7),
                                                        ("FASTSLAM1", "traj_fs1_loop2_N100_s7", "example_loop2", 7),
                                                        ("FASTSLAM2", "traj_fs2_loop902_N100_s3", "example_loop902", 3),
                                                        ("FASTSLAM1", "traj_fs1_loop902_N100_s3", "example_loop902", 3)])
def test_slam_backend_follows_the_reference_trajectory(tmp_path, method, golden, math, mapname, seed):
    g = load_golden(golden)
    maxsteps = 4000
    out, rows = run_backend(tmp_path, method, math, maxsteps=maxsteps, mapname=mapname, seed=seed)
    assert ("FastSLAM 2" if method == "FASTSLAM2" else "FastSLAM 1") in out and "control steps %d" % maxsteps in out
    assert rows.shape == (maxsteps, 8)
    # golden `ctl`[k] = control-step index (1-based iteration) at which observation step k happened
    ctl = g["ctl"]
    nobs = int((ctl <= maxsteps).sum())
    assert nobs >= 400
    est = rows[ctl[:nobs] - 1, 4:7]  # estimate logged at the control step of each observation
    true = rows[ctl[:nobs] - 1, 1:4]
    assert np.abs(true - g["true"][:nobs]).max() <= 1e-4  # the simulator front end reproduces the reference's true path
    d = np.hypot(est[:, 0] - g["est"][:nobs, 0], est[:, 1] - g["est"][:nobs, 1])
    first_res = int(np.argmax(g["resampled"][:nobs])) if g["resampled"][:nobs].any() else nobs
    assert d[:first_res].max() <= 1e-3, (first_res, d[:first_res].max())
    # (the first resampling step itself: a stratum on the other side of a cumulative-sum boundary picks a neighbouring ancestor)
    assert d[first_res:first_res + 1].max(initial=0.0) <= (1e-3 if math == "strict" and mapname == "example_webmap" else 1e-2), (first_res, d[first_res])
    # FastSLAM1's weights are well conditioned (GPU vs reference ~1e-4): ancestors stay identical for dozens of
    # resamples before one stratum lands on the other side of a cumulative-sum boundary
    # (strict build; the fast build's predict uses the bounded-angle polynomials, ~1e-6 per step on the pose, and a first
    # stratum changes sides within a dozen resamples: it is held to the first resample above and to the statistics below)
    if method == "FASTSLAM1" and math == "strict" and mapname == "example_webmap":
        assert d[:25].max() <= 1e-3, d[:25].max()
    err_g = np.hypot(est[:, 0] - true[:, 0], est[:, 1] - true[:, 1])
    err_r = np.hypot(g["est"][:nobs, 0] - g["true"][:nobs, 0], g["est"][:nobs, 1] - g["true"][:nobs, 1])
    # (one 100-particle run's mean error depends on which ancestors a few early resamples picked; on the loop maps -- slow vehicle,
    # 10 m sensor range, few landmarks in view -- two runs of the REFERENCE with different seeds differ by 2x)
    slack = (1.5, 0.05) if mapname == "example_webmap" else (2.5, 0.1)
    assert err_g.mean() <= slack[0] * err_r.mean() + slack[1], (err_g.mean(), err_r.mean())


def _philox_run(tmp_path, name, extra=(), n=4096):
    log = str(tmp_path / (name + ".csv"))
    r = subprocess.run([EXE, "-m", os.path.join(DATA, "example_webmap.mat"), "-method", "FASTSLAM2", "-NPARTICLES", str(n),
                        "-NEFFECTIVE", str(3 * n // 4), "-SWITCH_SEED_RANDOM", "7", "-log", log, *extra], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-800:] + r.stderr[-800:]
    return r.stdout, np.loadtxt(log, delimiter=",", skiprows=1)


def test_slam_backend_philox_full_run_tracks_the_true_path(tmp_path):
    """The throughput configuration of the binary (Philox noise, fast build) over the whole example_webmap run, in the three
    forms of its loop: the wrapper's loop call by call (-loop step: a predict launch and a synchronous estimate per control
    step), batched (default: one slamgpu_step per observation, estimates fetched 4 096 at a time) and batched with the
    observation made on the GPU (-observe device)."""
    out_s, rows_s = _philox_run(tmp_path, "step", ["-loop", "step"])
    assert rows_s.shape[0] > 17000 and "landmarks in map: 35" in out_s
    err_s = np.hypot(rows_s[:, 4] - rows_s[:, 1], rows_s[:, 5] - rows_s[:, 2])
    assert np.isfinite(err_s).all() and err_s.mean() < 1.0, err_s.mean()
    out_b, rows_b = _philox_run(tmp_path, "batched", ["-gpubusy", "1"])
    assert rows_b.shape[0] == 2172 and "landmarks in map: 35" in out_b and "wall time per observation step" in out_b and "GPU busy" in out_b
    # the same filter on the same tape: the batched loop folds the queued predicts into the update launch (float rounding
    # differs), so the two agree closely until a resample picks a different ancestor, and statistically afterwards
    by_iter = {int(r[0]): r for r in rows_s}
    ref = np.array([by_iter[int(r[0])][4:7] for r in rows_b])
    assert np.abs(rows_b[:8, 4:7] - ref[:8]).max() <= 1e-3
    err_b = np.hypot(rows_b[:, 4] - rows_b[:, 1], rows_b[:, 5] - rows_b[:, 2])
    # (one run's mean error depends on which ancestors a few early resamples picked: 0.35 m and 0.76 m seen for the two forms
    # at 4 096 particles, 0.69 m and 0.72 m at 100 000; the bound is the one the other whole-run tests use)
    assert np.isfinite(err_b).all() and err_b.mean() < 1.5, err_b.mean()
    out_d, rows_d = _philox_run(tmp_path, "device", ["-observe", "device"])
    assert rows_d.shape[0] == 2172 and "landmarks in map: 35" in out_d
    assert np.array_equal(rows_d[:, 1:4], rows_b[:, 1:4])  # the same true path
    err_d = np.hypot(rows_d[:, 4] - rows_d[:, 1], rows_d[:, 5] - rows_d[:, 2])
    assert np.isfinite(err_d).all() and err_d.mean() < 1.5, err_d.mean()  # (other sensor noise: Philox on the device)
    # refused combinations say why
    r = subprocess.run([EXE, "-m", os.path.join(DATA, "example_webmap.mat"), "-method", "FASTSLAM2", "-observe", "device", "-rng", "parity"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "-observe device needs" in r.stderr


def test_slam_backend_plot_stream_with_particles(tmp_path):
    """FASTSLAM2 on the GPU with -plot file:<frames>: per control step the wrapper's tail (fastslam2wrapper.cpp:92-117):
    loopTime, setParticles, setFeatureParticles (decimated by -plotstride), positions, laser lines, plot."""
    import struct
    frames = str(tmp_path / "fs2_frames.bin")
    r = subprocess.run([EXE, "-m", os.path.join(DATA, "example_webmap.mat"), "-method", "FASTSLAM2", "-NPARTICLES", "1000", "-NEFFECTIVE", "750",
                        "-SWITCH_SEED_RANDOM", "7", "-maxsteps", "60", "-plot", "file:" + frames, "-plotstride", "100"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-800:] + r.stderr[-800:]
    b = open(frames, "rb").read()
    off, msgs = 4, []
    for _ in range(struct.unpack_from("<I", b, 0)[0]):
        (nf,) = struct.unpack_from("<I", b, off)
        off += 4
        fr = []
        for _ in range(nf):
            (ln,) = struct.unpack_from("<I", b, off)
            off += 4
            fr.append(b[off:off + ln])
            off += ln
        msgs.append(fr)
    names = [m[0].decode() for m in msgs]
    per_step = ["loopTime", "setParticles", "setFeatureParticles", "addTruePosition", "addEstimatedPosition", "setCarTruePosition",
                "setCarEstimatedPosition", "setLaserLines", "plot"]
    body = names[11:-1]
    assert names[-1] == "endPlot" and len(body) == 60 * len(per_step) and body[-len(per_step):] == per_step
    last_particles = [m for m in msgs if m[0] == b"setParticles"][-1]
    assert struct.unpack(">i", last_particles[1])[0] == 10  # 1000 particles, every 100th
    last_features = [m for m in msgs if m[0] == b"setFeatureParticles"][-1]
    assert struct.unpack(">i", last_features[1])[0] == 10 * 6  # 6 landmarks in the map after 60 control steps
    xs = [struct.unpack(">d", f)[0] for f in last_particles[2:12]]
    assert all(np.isfinite(xs)) and 0.5 < np.mean(xs) < 10.0


@pytest.mark.parametrize("seed,math", [(8, "strict"), (7, "fast"), (9, "fast"), (11, "strict")])
def test_slam_backend_gated_association_builds_the_same_map(tmp_path, seed, math):
    """-assoc gated: no association table: every particle gates the observations against its own landmark estimates
    (slamgpu_associate: EKFSLAM::dataAssociate per particle, ekfslam.cpp:151-189), the weighted vote gives the step's labels, and
    the policy of host/gated.h turns them into the update's packet.  Round 5's policy (plurality vote, open on "new") was right
    decision by decision and wrong run by run: 22 of 40 whole runs ended at the map's capacity.  The cause was not the split votes
    the review suspected but UNANIMOUS ones (profiles/gated_association_whole_runs_r06.txt): closing the loop of example_webmap the
    landmarks of the first lap come back into view at 60 m with every particle ~1 m off along the line of sight -- ten standard
    deviations of the range sensor for a chi-square gate that knows only the particle's own landmark covariance -- so every
    particle says "new" and the first lap's map is duplicated; before that, the gates drop exactly the observations that contradict
    the estimate.  Round 6's second stage (a unique nearest mapped landmark in world coordinates explains what the gates leave
    unexplained; nothing is opened next to a mapped landmark) ends all 40 probed runs (seeds 7..16, both builds, 512 and 2 048
    particles) with EXACTLY the map's 35 landmarks, none retired, every decision the true one, and the mean position error of the
    same run with the reference's known association (0.49 against 0.54 m averaged over the 40; 35 of them under 1 m, the other five
    1.03-1.16 m with the known-association twin at 0.7-1.1 m: 512 particles and that seed's noise, not the association).
    Pinned here: seeds 7 and 9 in the fast build and 11 in the strict one -- three of round 5's runaways -- and round 5's good case;
    the map must come out exact and the error must be that of the known-association twin.  The reference has no FastSLAM version of
    this association (parity unpinned by nature); every single kernel decision is pinned to the reference's EKF gating
    (tests/test_association.py)."""
    import re

    def run(extra, name):
        log = str(tmp_path / name)
        r = subprocess.run([EXE, "-m", os.path.join(DATA, "example_webmap.mat"), "-method", "FASTSLAM2", "-NPARTICLES", "512", "-NEFFECTIVE", "384",
                            "-SWITCH_SEED_RANDOM", str(seed), "-math", math, "-log", log] + extra, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-800:] + r.stderr[-800:]
        rows = np.loadtxt(log, delimiter=",", skiprows=1)
        err = np.hypot(rows[:, 4] - rows[:, 1], rows[:, 5] - rows[:, 2])
        assert np.isfinite(err).all()
        return r.stdout, err
    out, err = run(["-assoc", "gated"], "gated.csv")
    m = re.search(r"landmarks in map: (\d+) \((\d+) opened, (\d+) retired", out)
    nl, opened, retired = int(m.group(1)), int(m.group(2)), int(m.group(3))
    assert (nl, opened, retired) == (35, 35, 0), out[-400:]          # the map, no duplicate ever opened, nothing to retire
    _, err_known = run(["-loop", "step"], "known.csv")              # dataAssociationKnown (core.cpp:91-120), same seed, same particles
    # (the twin differs by which observations the gates let through early on: individual runs land on either side of it -- over the 40
    # probed runs the gated mean is the smaller one)
    assert err.mean() < 1.0 and err.mean() <= 2.0 * err_known.mean() + 0.25, (err.mean(), err_known.mean())
    assert err.max() < 2.5, err.max()


def test_slam_backend_gated_association_on_the_largest_bundled_map(tmp_path):
    """... and on example_loop902 (117 landmarks on a loop driven twice, heading observed: the densest of the bundled maps, where the
    second stage's uniqueness tests have the least room): the map comes out exact and the error is the known-association twin's
    (all 18 probed runs on the three loop maps do: profiles/gated_association_whole_runs_r06.txt)."""
    import re

    def run(extra, name):
        log = str(tmp_path / name)
        r = subprocess.run([EXE, "-m", os.path.join(DATA, "example_loop902.mat"), "-method", "FASTSLAM2", "-NPARTICLES", "1024", "-NEFFECTIVE", "768",
                            "-SWITCH_SEED_RANDOM", "7", "-math", "fast", "-log", log] + extra, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-800:] + r.stderr[-800:]
        rows = np.loadtxt(log, delimiter=",", skiprows=1)
        return r.stdout, np.hypot(rows[:, 4] - rows[:, 1], rows[:, 5] - rows[:, 2])
    out, err = run(["-assoc", "gated"], "gated902.csv")
    m = re.search(r"landmarks in map: (\d+) \((\d+) opened, (\d+) retired", out)
    assert (int(m.group(1)), int(m.group(2)), int(m.group(3))) == (117, 117, 0), out[-400:]
    _, err_known = run(["-loop", "step"], "known902.csv")
    assert err.mean() < 0.5 and err.mean() <= 2.0 * err_known.mean() + 0.25 and err.max() < 1.5, (err.mean(), err_known.mean(), err.max())


def test_retired_landmarks_take_no_part_in_the_association():
    """slamgpu_retire_landmarks (round 6): a retired landmark is never a label again -- in the exhaustive scan and through the grid
    -- and the decisions about every other landmark are what they were."""
    import slam_amd as sg
    from slam_amd import host
    from conftest import sim_args
    tape = host.make_tape(sim_args("example_webmap", "FASTSLAM2", 1024, 7), max_obs=120)
    s = sg.SlamGpu(1024, tape["nlm"], method=2, n_effective=768, rng_mode=sg.RNG_PHILOX, seed=3, math_mode=1)
    for st in tape["steps"]:
        s.step(np.array(st["controls"], np.float32).reshape(-1, 3), tape["Q"], float(tape["dt"]), st["zf"], st["idf"], st["zn"], tape["R"])
    st = tape["steps"][-1]
    z = np.concatenate([st["zf"], st["zn"]]).reshape(-1, 2)
    assert len(st["idf"]) >= 3
    g1, g2 = float(tape["conf"].GATE_REJECT), float(tape["conf"].GATE_AUGMENT)
    for mode in (sg.capi.ASSOC_EXHAUSTIVE, sg.capi.ASSOC_GRID):
        lab0, cons0, _ = s.associate(z, tape["R"], g1, g2, mode=mode)[:3]
        found = [int(j) for j in st["idf"] if int(j) in set(int(c) for c in cons0)]
        assert len(found) >= 2   # (re-observed landmarks are found again)
    victim = found[0]
    s.retire_landmarks([victim])
    for mode in (sg.capi.ASSOC_EXHAUSTIVE, sg.capi.ASSOC_GRID):
        lab, cons, _ = s.associate(z, tape["R"], g1, g2, mode=mode)[:3]
        assert victim not in set(int(c) for c in cons) and not (np.asarray(lab) == victim).any()
        # the labels of every OTHER observation are untouched (the victim's own observation changes for everybody: particles it was
        # the nearest neighbour of, and particles for which it was the reason to discard rather than open -- `outer`, ekfslam.cpp:176)
        cols = np.asarray(cons0) != victim
        assert cols.sum() >= 2 and np.array_equal(np.asarray(lab)[:, cols], np.asarray(lab0)[:, cols])
    with pytest.raises(sg.SlamGpuError):
        s.retire_landmarks([s.nf()])
    # a new particle set is a new map: slamgpu_upload clears the marks
    s.upload(s.download())
    for mode in (sg.capi.ASSOC_EXHAUSTIVE, sg.capi.ASSOC_GRID):
        lab, cons, _ = s.associate(z, tape["R"], g1, g2, mode=mode)[:3]
        assert victim in set(int(c) for c in cons)
    s.close()


def test_slam_backend_gpus_k_is_independent_of_k(tmp_path):
    """slam-backend -gpus k (slamgpu_dist_group_*: one process, k shards; k above the device count = logical shards on
    device 0): the logged estimates of the whole run must not depend on k (Philox noise: same streams whatever the
    sharding) and must track the single-context loop."""
    def run(k, n=4096, maxsteps=6000):
        log = str(tmp_path / ("gpus%d.csv" % k))
        cmd = [EXE, "-m", os.path.join(DATA, "example_webmap.mat"), "-method", "FASTSLAM2", "-NPARTICLES", str(n), "-NEFFECTIVE", str(3 * n // 4),
               "-SWITCH_SEED_RANDOM", "7", "-log", log, "-maxsteps", str(maxsteps)] + (["-gpus", str(k)] if k else [])
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-800:] + r.stderr[-800:]
        return r.stdout, np.loadtxt(log, delimiter=",", skiprows=1)
    out1, single = run(0)
    by_iter = {int(r[0]): r for r in single}
    prev = None
    for k in (2, 4, 8):
        out, rows = run(k)
        assert "over %d logical shards" % k in out or "over %d GPUs" % k in out
        assert len(rows) >= 500
        # against the single-context loop: that one runs every predict as a launch of its own (it reports an estimate per
        # control step), this one folds the queued predicts into the update launch: same filter, float rounding differs, so
        # the runs agree closely until a resample picks a different ancestor, and statistically afterwards
        ref = np.array([by_iter[int(r[0])][4:7] for r in rows])
        assert np.abs(rows[:8, 4:7] - ref[:8]).max() <= 1e-3, k
        err_k = np.hypot(rows[:, 4] - rows[:, 1], rows[:, 5] - rows[:, 2]).mean()
        err_1 = np.hypot(ref[:, 0] - rows[:, 1], ref[:, 1] - rows[:, 2]).mean()
        assert err_k <= 1.5 * err_1 + 0.05, (k, err_k, err_1)
        if prev is not None:
            assert np.array_equal(prev[:, :7], rows[:, :7]), k
        prev = rows
    # not a multiple of 256 k: refused with the nearest valid size, nothing run
    r = subprocess.run([EXE, "-m", os.path.join(DATA, "example_webmap.mat"), "-method", "FASTSLAM2", "-NPARTICLES", "1000", "-gpus", "2"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "multiple of 512" in r.stderr


def test_slam_backend_gpus_k_fastslam1(tmp_path):
    """the same independence of k for FastSLAM1 (per-particle control noise drawn on the device: Philox keyed by the global
    particle id, so the streams do not depend on the sharding either)"""
    def run(k):
        log = str(tmp_path / ("fs1_gpus%d.csv" % k))
        cmd = [EXE, "-m", os.path.join(DATA, "example_webmap.mat"), "-method", "FASTSLAM1", "-NPARTICLES", "2048", "-NEFFECTIVE", "1536",
               "-SWITCH_SEED_RANDOM", "7", "-log", log, "-maxsteps", "4000", "-gpus", str(k)]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "FastSLAM 1" in r.stdout, r.stdout[-800:] + r.stderr[-800:]
        return np.loadtxt(log, delimiter=",", skiprows=1)
    a, b = run(2), run(4)
    assert len(a) >= 400 and np.array_equal(a[:, :7], b[:, :7])
    err = np.hypot(a[:, 4] - a[:, 1], a[:, 5] - a[:, 2])
    assert err.mean() < 1.5, err.mean()  # tracks the true path (FastSLAM1 at 2 048 particles: well under a metre and a half)


@pytest.mark.parametrize("seed,math,rng", [(7, "fast", "philox"), (11, "strict", "philox"), (9, "strict", "parity")])
def test_slam_backend_per_particle_association(tmp_path, seed, math, rng):
    """-assoc particle: the same gates, but nothing is reduced to a vote: every particle acts on its own decisions on a map of its
    own (slamgpu_update_particle; the kernel side is pinned in tests/test_gpu_particle_assoc.py).  The best particle's map must be
    the map -- 35 landmarks, at most one spurious, all 35 true ones found -- with at most a handful of extra slots ever opened by
    minorities, and the estimate must be as good as the known-association twin's; also with the reference's libc draws (-rng parity)."""
    import re

    def run(extra, name):
        log = str(tmp_path / name)
        r = subprocess.run([EXE, "-m", os.path.join(DATA, "example_webmap.mat"), "-method", "FASTSLAM2", "-NPARTICLES", "500", "-NEFFECTIVE", "375",
                            "-SWITCH_SEED_RANDOM", str(seed), "-math", math, "-rng", rng, "-log", log] + extra, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-800:] + r.stderr[-800:]
        rows = np.loadtxt(log, delimiter=",", skiprows=1)
        err = np.hypot(rows[:, 4] - rows[:, 1], rows[:, 5] - rows[:, 2])
        assert np.isfinite(err).all()
        return r.stdout, err
    out, err = run(["-assoc", "particle"], "particle.csv")
    m = re.search(r"landmarks in map: (\d+) \(the best particle's, number \d+; (\d+) of the 35 true landmarks within 1 m of one of them; (\d+) slots", out)
    assert m, out[-500:]
    held, covered, slots = int(m.group(1)), int(m.group(2)), int(m.group(3))
    _, err_known = run(["-loop", "step"], "known.csv")
    assert 35 <= held <= 36 and slots <= 40, out[-500:]
    assert err.mean() <= 1.2 * err_known.mean() + 0.25 and err.max() < 2.5, (err.mean(), err_known.mean(), err.max())
    if err_known.mean() < 0.5:
        assert covered >= 33, out[-500:]   # (a map that has not drifted as a whole covers the true landmarks)
