#!/usr/bin/env python3
"""GPU box check: large particle counts in one context (LDS for the block-total prefix grows with N/256)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import slam_amd
from slam_amd import host
for N in [int(a) for a in sys.argv[1:]] or [800768, 1000000, 2000000]:
    tape = host.make_tape(["-m", os.path.join(ROOT, "data", "example_webmap.mat"), "-method", "FASTSLAM2", "-NPARTICLES", N, "-NEFFECTIVE", int(0.75 * N), "-SWITCH_SEED_RANDOM", 7], max_obs=200)
    try:
        s = slam_amd.SlamGpu(N, tape["nlm"], method=2, n_effective=int(0.75 * N), rng_mode=slam_amd.RNG_PHILOX, seed=7, math_mode=1)
        t0 = time.perf_counter()
        for st in tape["steps"]:
            s.step(np.array(st["controls"], np.float32).reshape(-1, 3), tape["Q"], float(tape["dt"]), st["zf"], st["idf"], st["zn"], tape["R"])
        est, ne, rs = s.history_fetch()
        dt = time.perf_counter() - t0
        err = np.mean([np.hypot(e[0] - st["true"][0], e[1] - st["true"][1]) for e, st in zip(est, tape["steps"])])
        print("N %8d: %.1f us/step, %.3g particle-updates/s, resampled %d/%d, mean pose error %.3f m" % (N, 1e6 * dt / len(est), N * len(est) / dt, rs.sum(), len(rs), err))
        s.close()
    except Exception as e:
        print("N %8d: FAILED %s" % (N, e))
