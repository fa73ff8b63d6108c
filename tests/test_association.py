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
