#!/usr/bin/env python3
"""GPU box: how far (in 256-particle blocks) a stratified ancestor lies from its offspring, N = 100 000, example_webmap."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import slam_amd as sg  # noqa: E402
from slam_amd import host  # noqa: E402

N = 100000
tp = host.make_tape(["-m", os.path.join(ROOT, "data", "example_webmap.mat"), "-method", "FASTSLAM2", "-NPARTICLES", N, "-NEFFECTIVE", int(0.75 * N),
                     "-SWITCH_SEED_RANDOM", 7], max_obs=1300)
s = sg.SlamGpu(N, tp["nlm"], method=2, n_effective=int(0.75 * N), rng_mode=sg.RNG_PHILOX, seed=7, math_mode=1)
hist = np.zeros(64, np.int64)
nres = 0
for k, st in enumerate(tp["steps"]):
    s.step(np.array(st["controls"], np.float32).reshape(-1, 3), tp["Q"], float(tp["dt"]), st["zf"], st["idf"], st["zn"], tp["R"])
    if k >= 1000:
        ne, did, _ = s.stats()
        if did:
            a = s.ancestors().astype(np.int64)
            d = np.abs(a // 256 - np.arange(N) // 256)
            hist += np.bincount(np.minimum(d, 63), minlength=64)
            nres += 1
s.close()
tot = hist.sum()
print("resamples sampled: %d" % nres)
for r in (0, 1, 2, 3, 4, 8, 16):
    print("ancestor within +-%d blocks: %.2f %%" % (r, 100.0 * hist[:r + 1].sum() / tot))
