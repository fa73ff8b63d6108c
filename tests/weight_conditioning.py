#!/usr/bin/env python3
"""How sharply do the reference's own FastSLAM2 weights react to rounding-level changes of their inputs?

Runs the oracle (bit-exact restatement of the reference) on the golden pre-update particle sets twice: once as
recorded, once with every stored float of the pose covariance Pv nudged by one ulp in a random direction (the pose,
the map and the random draws stay identical).  The relative change of the resulting weights is the noise floor any
implementation that does not replay the reference's float operations one by one has to live with: FastSLAM2 inverts
the predicted Pv, which is close to rank 2 (eight rank-2 process-noise increments with nearly parallel directions).
CPU only; uses the oracle as the thing measured, so this is a diagnostic, not a product path."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import orc
from conftest import load_golden

f32 = np.float32


def nudge(a, rng):
    up = rng.integers(0, 2, a.shape).astype(bool)
    return np.where(up, np.nextafter(a, f32(np.inf)), np.nextafter(a, f32(-np.inf))).astype(f32)


def measure(name, what=("Pv",), seed=0):
    o = orc.Oracle()
    g = load_golden(name)
    rng = np.random.default_rng(seed)
    rel = []
    for k in g["snap_steps"]:
        m, n = g["m"][k - 1], g["n"][k - 1]
        if m == 0:
            continue
        out = []
        for pert in (False, True):
            pre = {key: np.array(g["snap%d_pre_%s" % (k, key)]) for key in ("xv", "Pv", "w", "xf", "Pf")}
            if pert:
                for key in what:
                    pre[key] = nudge(pre[key], rng)
            N = pre["w"].shape[0]
            pre["nf"] = pre["xf"].shape[1]
            P = o.particles(N, 64)
            P.set(pre)
            P.update_local(ALGO(g), g["zf"][k - 1, :m], g["idf"][k - 1, :m], g["zn"][k - 1, :n], g["meta_R"],
                           np.ascontiguousarray(g["snap%d_normals" % k]))
            out.append(P.get()["w"].astype(np.float64))
            P.close()
        ok = np.isfinite(out[0]) & np.isfinite(out[1]) & (out[0] > 0)
        rel.append(np.abs(out[1][ok] / out[0][ok] - 1.0))
    rel = np.concatenate(rel)
    return dict(median=float(np.median(rel)), p99=float(np.quantile(rel, 0.99)), max=float(rel.max()), count=int(rel.size))


def ALGO(g):
    return orc.Algo(2, int(g["meta_use_heading"]), int(g["meta_add_predict_noise"]), int(g["meta_resample"]),
                    int(g["meta_n_effective"]), float(g["meta_wheel_base"]), float(g["meta_sigma_phi"]))


if __name__ == "__main__":
    for name in ("traj_fs2_webmap_N100_s7", "traj_fs2_webmap_N1000_s1"):
        for what in (("Pv",), ("xv",), ("Pf",)):
            print(name, "1-ulp nudge of", what, measure(name, what))
