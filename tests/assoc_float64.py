"""Float64 restatement of per-particle gated nearest-neighbour association (checker for slamgpu_associate): the reference's
EKFSLAM::dataAssociate / ekfComputeAssociation (algorithms/ekfslam.cpp:131-189) for a particle whose pose is known, so that
S = Hf Pf Hf^T + R.  PARITY: pinned to the reference's own function on tests/golden/kat_assoc.npz (decisions); the FastSLAM
use of it is the build's extension (the reference has no per-particle association)."""
import numpy as np


def wrap(a):
    return (a + np.pi) % (2 * np.pi) - np.pi


def associate(xv, xf, Pf, z, R, gate1, gate2, want_margin=False):
    """xv[3], xf[nf,2], Pf[nf,2,2], z[nz,2] -> labels[nz] (landmark, -1 new, -2 dropped) [+ margin: how close the decision was]"""
    xv, xf, Pf, z, R = (np.asarray(a, np.float64) for a in (xv, xf, Pf, z, R))
    nf, nz = xf.shape[0], z.shape[0]
    labels = np.zeros(nz, np.int32)
    margin = np.full(nz, np.inf)
    for q in range(nz):
        jbest, nbest, outer = -1, np.inf, np.inf
        nds = []
        for j in range(nf):
            dx, dy = xf[j] - xv[:2]
            d2 = dx * dx + dy * dy
            d = np.sqrt(d2)
            zp = np.array([d, np.arctan2(dy, dx) - xv[2]])
            Hf = np.array([[dx / d, dy / d], [-dy / d2, dx / d2]])
            S = Hf @ Pf[j] @ Hf.T + R
            v = z[q] - zp
            v[1] = wrap(v[1])
            nis = float(v @ np.linalg.solve(S, v))
            nd = nis + np.log(np.linalg.det(S))
            margin[q] = min(margin[q], abs(nis - gate1), abs(nis - gate2))
            if nis < gate1 and nd < nbest:
                nds.append(nd)
                nbest, jbest = nd, j
            elif nis < outer:
                outer = nis
        if len(nds) > 1:
            s_ = np.sort(nds)
            margin[q] = min(margin[q], s_[1] - s_[0])
        labels[q] = jbest if jbest > -1 else (-1 if outer > gate2 else -2)
    return (labels, margin) if want_margin else labels
