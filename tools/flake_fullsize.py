#!/usr/bin/env python3
"""GPU box diagnostic for tests/test_gpu_parity.py::test_full_size_philox_vs_oracle: the same drive, printing per resampling step
the share of particles whose pose differs from the oracle's, where they sit, and how far the GPU's ancestors are from the oracle's."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import slam_amd as sg  # noqa: E402
from conftest import sim_args  # noqa: E402
from oracle import orc  # noqa: E402

MODE = int(sys.argv[1]) if len(sys.argv) > 1 else 0
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 8
N = 100000
orc.build_oracle()
oracle = orc.Oracle()
o = oracle.sim(sim_args("example_webmap", "FASTSLAM2", N, 7))
o.set_rng(1, 7)
algo = o.algo()
Q, R, dt = o.noise()
s = sg.SlamGpu(N, o.nlm, method=2, n_effective=algo.n_effective, wheel_base=algo.wheel_base, rng_mode=sg.RNG_PHILOX, seed=7, math_mode=MODE)
k = 0
while k < STEPS:
    a = o.control()
    x, vg = o.true_pose()
    s.predict(float(vg[0]), float(vg[1]), Q, float(dt), float(x[2]))
    if a == 1:
        o.observe()
        ob = o.last_obs()
        s.update(ob["zf"], ob["idf"], ob["zn"], R)
        k += 1
        got, exp = s.download(), o.particles()
        ne_o, did_o = o.last_resample()
        ne_g, did_g, wsum = s.stats()
        line = "step %d m=%d n=%d neff gpu %.1f oracle %.1f did %d/%d" % (k, len(ob["idf"]), ob["zn"].shape[0], ne_g, ne_o, did_g, did_o)
        if did_g:
            keep = s.ancestors()
            ko = o.last_keep() if hasattr(o, "last_keep") else None
            bad = np.abs(got["xv"] - exp["xv"]).max(axis=1) > 2e-4
            idx = np.flatnonzero(bad)
            line += " | bad %.4f (%d) first %s last %s" % (bad.mean(), idx.size, idx[:4], idx[-4:])
            if idx.size:
                h, _ = np.histogram(idx, bins=10, range=(0, N))
                line += " per-decile %s" % list(h)
            if ko is not None:
                d = np.abs(keep.astype(np.int64) - ko.astype(np.int64))
                line += " | ancestors differ %.4f, >1: %.4f, max %d" % ((d > 0).mean(), (d > 1).mean(), d.max())
        else:
            line += " | max pose diff %.2e" % np.abs(got["xv"] - exp["xv"]).max()
        print(line, flush=True)
