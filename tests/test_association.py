"""Per-particle gated nearest-neighbour association.  The reference implements it for EKF-SLAM
only (EKFSLAM::dataAssociate, algorithms/ekfslam.cpp:151-189); tests/golden/kat_assoc.npz holds the decisions of THAT
function, run from the reference's own objects on single FastSLAM particles (pose known => P = blockdiag(0, Pf_j)).
  * CPU: the float64 restatement (tests/assoc_float64.py) reproduces every decision;
  * GPU (-m gpu): slamgpu_associate (float32, one particle per work-item) reproduces them except where a gate or a tie is
    closer than float32 can resolve (margin of the float64 evaluation below 1e-3: such cases are skipped, and must be rare),
    and the weighted consensus is the majority label."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

f32 = np.float32


@pytest.fixture(scope="module")
def kat_assoc():
    return np.load(os.path.join(GOLDEN, "kat_assoc.npz"))


def test_float64_restatement_matches_the_reference_decisions(kat_assoc):
    import assoc_float64 as A
    g1, g2 = (float(x) for x in kat_assoc["gates"])
    R = np.array([[0.1 ** 2, 0], [0, 0.017453292519943 ** 2]], f32)
    tot = 0
    kinds = {0: 0, -1: 0, -2: 0}
    for g in range(int(kat_assoc["n_groups"])):
        xv, xf, Pf, z, lab = (kat_assoc["g%d_%s" % (g, k)] for k in ("xv", "xf", "Pf", "z", "lab"))
        for i in range(xv.shape[0]):
            got = A.associate(xv[i], xf[i], Pf[i], z, R, g1, g2)
            assert np.array_equal(got, lab[i]), (g, i, got, lab[i])
            tot += got.size
            for l in lab[i]:
                kinds[min(int(l), 0)] += 1
    assert tot == 10 * 16 * 11 and min(kinds.values()) >= 100  # all three outcomes well represented


@pytest.mark.gpu
@pytest.mark.parametrize("math_mode", [0, 1], ids=["strict", "fast"])
def test_device_association_matches_the_reference_decisions(kat_assoc, math_mode):
    import slam_amd as sg
    import assoc_float64 as A
    g1, g2 = (float(x) for x in kat_assoc["gates"])
    R = np.array([[0.1 ** 2, 0], [0, 0.017453292519943 ** 2]], f32)
    checked = skipped = 0
    for g in range(int(kat_assoc["n_groups"])):
        xv, xf, Pf, z, lab = (kat_assoc["g%d_%s" % (g, k)] for k in ("xv", "xf", "Pf", "z", "lab"))
        N, nf = xv.shape[0], xf.shape[1]
        s = sg.SlamGpu(N, 32, method=2, rng_mode=sg.RNG_PHILOX, math_mode=math_mode)
        w = np.full(N, 1.0 / N, f32)
        s.upload(dict(nf=nf, xv=xv, Pv=np.zeros((N, 3, 3), f32), w=w, xf=xf, Pf=Pf))
        got, cons, sup = s.associate(z, R, g1, g2)
        s.close()
        assert got.shape == lab.shape
        for i in range(N):
            _, margin = A.associate(xv[i], xf[i], Pf[i], z, R, g1, g2, want_margin=True)
            clear = margin > 1e-3
            assert np.array_equal(got[i][clear], lab[i][clear]), (g, i, got[i], lab[i], margin)
            checked += int(clear.sum())
            skipped += int((~clear).sum())
        # consensus = the label with the largest weight share (equal weights here: the mode), duplicates resolved
        for q in range(z.shape[0]):
            vals, counts = np.unique(got[:, q], return_counts=True)
            if cons[q] != sg.capi.ASSOC_DISCARD or vals[np.argmax(counts)] == sg.capi.ASSOC_DISCARD:
                assert counts[list(vals).index(cons[q])] == counts.max() or cons[q] == sg.capi.ASSOC_DISCARD
            assert 0.0 < sup[q] <= 1.0 + 1e-6
        pos = cons[cons >= 0]
        assert len(set(pos.tolist())) == len(pos)  # one observation per landmark
    assert checked >= 1600 and skipped <= 40, (checked, skipped)


@pytest.mark.gpu
@pytest.mark.parametrize("prefilter", ["lists", "grid"])
@pytest.mark.parametrize("math_mode", [0, 1], ids=["strict", "fast"])
def test_grid_prefilter_never_changes_a_reference_decision(kat_assoc, math_mode, prefilter, monkeypatch):
    """slamgpu_associate_ex, SLAMGPU_ASSOC_GRID against SLAMGPU_ASSOC_EXHAUSTIVE on the reference's decision vectors: the
    prefilter skips a landmark only where a bound that holds for every particle rules both gates out, so the label arrays must
    be identical -- every group, every particle, every observation, borderline cases included.  Both forms of the prefilter: one
    candidate list per observation (round 6: what a call with at most 4 096 observations takes) and the uniform grid."""
    import slam_amd as sg
    if prefilter == "grid":
        monkeypatch.setenv("SLAMGPU_NO_ASSOC_LISTS", "1")
    g1, g2 = (float(x) for x in kat_assoc["gates"])
    R = np.array([[0.1 ** 2, 0], [0, 0.017453292519943 ** 2]], f32)
    same = 0
    for g in range(int(kat_assoc["n_groups"])):
        xv, xf, Pf, z = (kat_assoc["g%d_%s" % (g, k)] for k in ("xv", "xf", "Pf", "z"))
        N, nf = xv.shape[0], xf.shape[1]
        s = sg.SlamGpu(N, 32, method=2, rng_mode=sg.RNG_PHILOX, math_mode=math_mode)
        s.upload(dict(nf=nf, xv=xv, Pv=np.zeros((N, 3, 3), f32), w=np.full(N, 1.0 / N, f32), xf=xf, Pf=Pf))
        a, ca, _ = s.associate(z, R, g1, g2, mode=sg.capi.ASSOC_EXHAUSTIVE)
        b, cb, _, st = s.associate(z, R, g1, g2, mode=sg.capi.ASSOC_GRID, want_stats=True)
        s.close()
        assert st["grid"] and np.array_equal(a, b) and np.array_equal(ca, cb), g
        same += a.size
    assert same == 10 * 16 * 11


@pytest.mark.gpu
@pytest.mark.parametrize("prefilter", ["lists", "grid", "lists-overflow"])
def test_grid_prefilter_on_a_running_filter(tmp_path, prefilter, monkeypatch):
    """The same identity on the state of a running filter, where it is meant to pay: 4 096 particles on a synthetic
    1 000-landmark map (MAX_RANGE 30), the step's real observations plus shifted ones (new-landmark and discard outcomes);
    the grid evaluates a small fraction of the N * nz * Nf triples of the exhaustive scan.  Three ways: the candidate lists, the grid, and
    lists made too short on purpose (the call must fall back to the grid by itself)."""
    import slam_amd as sg
    from conftest import DATA, sim_args
    from slam_amd import host
    if prefilter == "grid":
        monkeypatch.setenv("SLAMGPU_NO_ASSOC_LISTS", "1")
    if prefilter == "lists-overflow":  # lists of ONE entry overflow at once: the call must notice and take the grid, same labels
        monkeypatch.setenv("SLAMGPU_ASSOC_LCAP", "1")
    lm = host.synthetic_landmarks(777, 1000, -130, 100, -100, 90)
    h0 = host.HostSim(sim_args("example_webmap", "FASTSLAM2", 100, 7))
    _, wp = h0.map()
    h0.close()
    mp = str(tmp_path / "syn1000.mat")
    host.write_map(mp, lm, wp)
    open(str(tmp_path / "syn1000.ini"), "w").write(open(os.path.join(DATA, "example_webmap.ini")).read().replace(
        "MAX_RANGE           = 60.0", "MAX_RANGE           = 30.0"))
    N = 4096
    tape = host.make_tape(["-m", mp, "-method", "FASTSLAM2", "-SWITCH_SEED_RANDOM", 3], max_obs=150)
    s = sg.SlamGpu(N, tape["nlm"], method=2, n_effective=int(0.75 * N), rng_mode=sg.RNG_PHILOX, seed=9, math_mode=1)
    R = tape["R"]
    rng = np.random.default_rng(5)
    checked = kinds = 0
    for i, st in enumerate(tape["steps"]):
        if i % 25 == 24:
            z = np.concatenate([st["zf"], st["zn"]]).reshape(-1, 2)
            z2 = z.copy()
            z2[:, 0] += rng.normal(0, 0.4, z.shape[0]).astype(f32)   # some of these fall between the gates or outside both
            zz = np.concatenate([z, z2])
            a, ca, _, sa = s.associate(zz, R, 4.0, 25.0, mode=sg.capi.ASSOC_EXHAUSTIVE, want_stats=True)
            b, cb, _, sb = s.associate(zz, R, 4.0, 25.0, mode=sg.capi.ASSOC_GRID, want_stats=True)
            assert sb["grid"] and np.array_equal(a, b) and np.array_equal(ca, cb), i
            assert sb["triples"] < 0.1 * sa["triples"], (i, sb, sa)
            # (entries: a list holds a handful per observation, the grid hundreds per cell: which form answered shows here)
            assert (sb["grid_entries"] < 40 * zz.shape[0]) == (prefilter == "lists"), (i, prefilter, sb)
            checked += a.size
            kinds |= (1 if (a >= 0).any() else 0) | (2 if (a == -1).any() else 0) | (4 if (a == -2).any() else 0)
        s.step(np.array(st["controls"], f32).reshape(-1, 3), tape["Q"], float(tape["dt"]), st["zf"], st["idf"], st["zn"], R)
    s.close()
    assert checked > 1e6 and kinds == 7
