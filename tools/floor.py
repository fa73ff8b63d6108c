#!/usr/bin/env python3
"""GPU box diagnostic: per-step time of the one-launch pipeline for a tiny context (latency floor), with and without
observations."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import slam_amd
from slam_amd import host
f32 = np.float32
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
tape = host.make_tape(["-m", os.path.join(ROOT, "data", "example_webmap.mat"), "-method", "FASTSLAM2", "-NPARTICLES", N, "-NEFFECTIVE", int(0.75 * N), "-SWITCH_SEED_RANDOM", 7], max_obs=1200)
Q, R, dt = tape["Q"], tape["R"], float(tape["dt"])
for label in ("no observations, 1 predict", "no observations, 8 predicts", "webmap tape"):
    s = slam_amd.SlamGpu(N, tape["nlm"], method=2, n_effective=int(0.75 * N), rng_mode=slam_amd.RNG_PHILOX, seed=7, math_mode=1)
    e2 = np.zeros((0, 2), f32); ei = np.zeros(0, np.int32)
    if label.startswith("no obs"):
        k = 1 if "1 predict" in label else 8
        calls = [s.prepare_step(np.array([[3.0, 0.0, 0.0]] * k, f32), Q, dt, e2, ei, e2, R) for _ in range(1200)]
    else:
        calls = [s.prepare_step(np.array(st["controls"], f32).reshape(-1, 3), Q, dt, st["zf"], st["idf"], st["zn"], R) for st in tape["steps"]]
    for c in calls[:100]:
        c()
    s.sync(); s.estimate_fetch()
    s.timer_start()
    for c in calls[100:1100]:
        c()
    ms = s.timer_stop()
    print("N %d  %-28s %.2f us/step (device time)" % (N, label, ms))
    s.close()
