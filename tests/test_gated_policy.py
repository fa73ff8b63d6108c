"""The policy between the gated association's vote and the update's packet (slam_amd/csrc/host/gated.h through the C ABI of
libslamhost: slamhost_gated_*), on hand-made cases: no GPU.  The labels it consumes are EKFSLAM::dataAssociate's per particle
(ekfslam.cpp:151-189; tests/test_association.py pins those); what is checked here is what round 6 added on top of the vote --
supermajority, the second stage in world coordinates with its uniqueness tests, the landmark credits."""
import numpy as np
import pytest

NEW, DISCARD = -1, -2
f32 = np.float32


@pytest.fixture()
def host():
    from slam_amd import host as h
    return h


def step(pol, z, cons, sup, xf, xv=(0.0, 0.0, 0.0), max_range=60.0, room=10):
    zf, idf, zn, ret = pol.step(np.asarray(z, f32), cons, sup, xv, np.asarray(xf, f32).reshape(-1, 2), max_range, room)
    return [tuple(np.round(r, 3)) for r in zf], list(idf), [tuple(np.round(r, 3)) for r in zn], list(ret)


def test_votes_become_a_packet_only_with_enough_weight_behind_them(host):
    pol = host.GatedPolicy(rescue=0)
    xf = [[10.0, 0.0], [0.0, 20.0]]
    z = [[10.0, 0.0], [20.0, np.pi / 2], [30.0, 1.0], [40.0, -1.0]]
    # landmark 0 with 60 % of the weight: matched; landmark 1 with 40 %: not; "new" with 95 %: opened; "new" with 80 %: not
    zf, idf, zn, ret = step(pol, z, [0, 1, NEW, NEW], [0.6, 0.4, 0.95, 0.8], xf)
    assert idf == [0] and len(zn) == 1 and abs(zn[0][0] - 30.0) < 1e-6 and ret == []
    c = pol.counts()
    assert c["opened"] == 1 and c["unused"] == 2 and c["second_stage_matches"] == 0
    # round 5's policy (enabled = 0): every plurality label is taken
    old = host.GatedPolicy(enabled=0)
    _, idf, zn, _ = step(old, z, [0, 1, NEW, NEW], [0.6, 0.4, 0.95, 0.8], xf)
    assert idf == [0, 1] and len(zn) == 2


def test_second_stage_matches_a_unique_nearby_landmark_and_opens_nothing_beside_it(host):
    pol = host.GatedPolicy()
    xf = [[30.0, 0.0], [0.0, 50.0]]
    # the loop closes: every particle says "new" for an observation that lands 1.2 m from landmark 0 (radius at 31.2 m: 2 + 5 % = 3.56 m)
    zf, idf, zn, _ = step(pol, [[31.2, 0.0]], [NEW], [1.0], xf)
    assert idf == [0] and zn == [] and pol.counts()["second_stage_matches"] == 1
    # between one and two radii of exactly one mapped landmark: neither matched nor opened
    zf, idf, zn, _ = step(pol, [[35.5, 0.0]], [NEW], [1.0], xf)
    assert idf == [] and zn == [] and pol.counts()["refused_new"] == 1
    # beyond two radii of everything: a new landmark
    zf, idf, zn, _ = step(pol, [[45.0, 0.0]], [NEW], [1.0], xf)
    assert idf == [] and len(zn) == 1
    # a discarded observation (between the gates for most particles) is rescued the same way
    zf, idf, zn, _ = step(pol, [[29.0, 0.02]], [DISCARD], [0.7], xf)
    assert idf == [0]


def test_second_stage_steps_back_where_the_surroundings_are_dense(host):
    pol = host.GatedPolicy()
    # two mapped landmarks 2 m apart: the geometry cannot say which one a drifted observation belongs to, nor that it is not a third
    xf = [[30.0, 0.0], [30.0, 2.0]]
    zf, idf, zn, _ = step(pol, [[30.5, 0.03]], [NEW], [1.0], xf)
    assert idf == [] and len(zn) == 1            # the gates' verdict stands
    # ... and two of this step's observations next to ONE mapped landmark (an unmapped neighbour may be there): the same
    pol2 = host.GatedPolicy()
    zf, idf, zn, _ = step(pol2, [[30.5, 0.0], [30.8, 0.05]], [NEW, NEW], [1.0, 1.0], [[30.0, 0.0]])
    assert idf == [] and len(zn) == 2
    # one landmark per observation and per step: two unexplained observations never share a rescued landmark
    pol3 = host.GatedPolicy(unique_ratio=1.0)
    zf, idf, zn, _ = step(pol3, [[30.5, 0.0], [10.0, 2.0]], [DISCARD, DISCARD], [1.0, 1.0], [[30.0, 0.0]])
    assert idf == [0]


def test_credits_retire_a_landmark_that_is_in_view_and_never_matched(host):
    pol = host.GatedPolicy(rescue=0, retire_below=-2, credit_start=1)
    xf = [[20.0, 0.0], [-20.0, 0.0]]     # landmark 0 ahead of the vehicle, landmark 1 behind it (out of the sensor's half disc)
    retired = []
    for k in range(6):
        _, idf, _, ret = step(pol, [[20.0, 0.0]], [DISCARD], [1.0], xf)
        retired += ret
        assert idf == []
    assert retired == [0]                # 1 -> 0 -> -1 -> -2 -> -3 < -2: retired on the fourth miss; landmark 1 was never expected
    c = pol.counts()
    assert c["retired"] == 1 and c["in_use"] == 1
    # a retired landmark is not matched again even if the vote names it
    _, idf, _, _ = step(pol, [[20.0, 0.0]], [0], [1.0], xf)
    assert idf == []
    # a landmark that is matched more often than it is missed keeps its credit (one that is matched a third of the time it is in view
    # does not: +1, -1, -1 per three steps)
    pol2 = host.GatedPolicy(rescue=0, retire_below=-2)
    for k in range(30):
        lab = DISCARD if k % 3 == 2 else 0
        _, _, _, ret = step(pol2, [[20.0, 0.0]], [lab], [1.0], xf)
        assert ret == []


def test_unknown_tunable_is_refused(host):
    with pytest.raises(ValueError):
        host.GatedPolicy(no_such_knob=1)


@pytest.mark.parametrize("seed,nf,nz,span", [(1, 2000, 300, 120.0), (2, 400, 60, 60.0), (3, 9000, 800, 250.0), (4, 35, 9, 200.0)])
def test_second_stage_grid_gives_the_decisions_of_the_full_scan(host, seed, nf, nz, span):
    """The second stage searches a uniform grid of the mapped landmarks and of the step's points (end of round 6: every pending
    observation scanning all of them was 2.7 ms of a step on the 10 000-landmark map).  Its tests only compare distances with bounds
    no larger than a cell, so the decisions must be those of the full scan (tunable grid = 0: one cell): random maps from dense (2 m
    between landmarks) to sparse, observations placed on, near and away from landmarks, votes that leave everything to the stage."""
    rng = np.random.default_rng(seed)
    xf = (rng.random((nf, 2)) * span - span / 2).astype(f32)
    xv = np.array([1.0, -2.0, 0.3], f32)
    # observations: a third on a landmark (+ noise of a metre), a third a few metres off one, a third anywhere
    tgt = xf[rng.integers(0, nf, nz)] + np.where(np.arange(nz)[:, None] % 3 == 0, rng.normal(0, 0.7, (nz, 2)),
                                                 np.where(np.arange(nz)[:, None] % 3 == 1, rng.normal(0, 4.0, (nz, 2)), rng.normal(0, span / 3, (nz, 2))))
    dx, dy = tgt[:, 0] - xv[0], tgt[:, 1] - xv[1]
    z = np.stack([np.hypot(dx, dy), np.arctan2(dy, dx) - xv[2]], axis=1).astype(f32)
    cons = np.where(rng.random(nz) < 0.5, NEW, DISCARD).astype(np.int32)
    sup = rng.uniform(0.5, 1.0, nz).astype(f32)
    outs = []
    for grid in (1, 0):
        pol = host.GatedPolicy(grid=grid)
        outs.append((pol.step(z, cons, sup, xv, xf, 300.0, 10 ** 6), pol.counts()))
    (a, ca), (b, cb) = outs
    for x, y in zip(a, b):
        assert np.array_equal(np.asarray(x), np.asarray(y))
    assert ca == cb and ca["second_stage_matches"] + ca["opened"] > 0
