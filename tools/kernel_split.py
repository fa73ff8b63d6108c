#!/usr/bin/env python3
"""GPU box diagnostic: per-kernel HIP-event times with the predicts forced into their own launch (estimate between
predicts and update), to see how the fused update launch splits into predict and update work."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import slam_amd
from slam_amd import host
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
tape = host.make_tape(["-m", os.path.join(ROOT, "data", "example_webmap.mat"), "-method", "FASTSLAM2", "-NPARTICLES", N, "-NEFFECTIVE", int(0.75 * N), "-SWITCH_SEED_RANDOM", 7], max_obs=700)
for mode in ("fused", "split"):
    s = slam_amd.SlamGpu(N, tape["nlm"], method=2, n_effective=int(0.75 * N), rng_mode=slam_amd.RNG_PHILOX, seed=7)
    for k, st in enumerate(tape["steps"]):
        if k == 100:
            s.sync(); s.profile(True)
        for (V, G, phi) in st["controls"]:
            s.predict(V, G, tape["Q"], float(tape["dt"]), phi)
        if mode == "split":
            s.sync()  # flushes the queued predicts as their own launch
        s.update(st["zf"], st["idf"], st["zn"], tape["R"])
    out = {k: s.kernel_time(k) for k in ("fs2_update", "resample", "predict")}
    print(N, mode, {k: "%.1f us x %d" % (1e3 * v[0] / max(v[1], 1), v[1]) for k, v in out.items()})
    s.close()
