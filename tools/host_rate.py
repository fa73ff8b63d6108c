#!/usr/bin/env python3
"""GPU box diagnostic: is the step loop bound by the host (enqueue rate) or by the GPU?  Times the enqueue loop
alone (before the final sync) and the whole run, for a few particle counts.
usage: python tools/host_rate.py [fast|strict] [torch]     (torch: import torch first, i.e. run on ITS bundled HIP runtime,
as the multi-GPU bench must: measured 9.8 us of host time per step against 4.9 us on /opt/rocm's runtime)"""
import os, sys, time
import numpy as np
if "torch" in sys.argv[1:]:
    import torch  # noqa: F401
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import slam_amd
from slam_amd import host
mm = slam_amd.MATH_STRICT if "strict" in sys.argv[1:] else slam_amd.MATH_FAST
for N in (1024, 100000, 400000):
    tape = host.make_tape(["-m", os.path.join(ROOT, "data", "example_webmap.mat"), "-method", "FASTSLAM2", "-NPARTICLES", N, "-NEFFECTIVE", int(0.75 * N), "-SWITCH_SEED_RANDOM", 7], max_obs=1300)
    s = slam_amd.SlamGpu(N, tape["nlm"], method=2, n_effective=int(0.75 * N), rng_mode=slam_amd.RNG_PHILOX, seed=7, math_mode=mm)
    calls = [s.prepare_step(np.array(st["controls"], np.float32).reshape(-1, 3), tape["Q"], float(tape["dt"]), st["zf"], st["idf"], st["zn"], tape["R"]) for st in tape["steps"]]
    for c in calls[:100]:
        c()
    s.sync(); s.estimate_fetch()
    t0 = time.perf_counter()
    for c in calls[100:1300]:
        c()
    t1 = time.perf_counter()
    s.sync()
    t2 = time.perf_counter()
    print("N %7d  enqueue %.2f us/step  total %.2f us/step" % (N, 1e6 * (t1 - t0) / 1200, 1e6 * (t2 - t0) / 1200))
    s.close()
