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
