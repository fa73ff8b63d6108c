#!/usr/bin/env python3
"""sha256 of a run's history and final state (FASTSLAM2, example_webmap, 4 096 particles, 400 observation steps, Philox) under
the library SLAMGPU_LIB names: two builds that must agree bit for bit print the same hash (round 6: the packed-FP32 variant of
the strict build against the shipped one).  usage: state_hash.py [strict|fast]"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import slam_amd  # noqa: E402
from slam_amd import host  # noqa: E402

math = slam_amd.MATH_STRICT if (len(sys.argv) < 2 or sys.argv[1] == "strict") else slam_amd.MATH_FAST
N = 4096
tape = host.make_tape(["-m", os.path.join(ROOT, "data", "example_webmap.mat"), "-method", "FASTSLAM2", "-NPARTICLES", N, "-NEFFECTIVE", int(0.75 * N),
                       "-SWITCH_SEED_RANDOM", 7], max_obs=400)
s = slam_amd.SlamGpu(N, tape["nlm"], method=2, n_effective=int(0.75 * N), rng_mode=slam_amd.RNG_PHILOX, seed=7, math_mode=math)
for st in tape["steps"]:
    s.step(np.array(st["controls"], np.float32).reshape(-1, 3), tape["Q"], float(tape["dt"]), st["zf"], st["idf"], st["zn"], tape["R"])
h = hashlib.sha256()
for a in s.history_fetch():
    h.update(np.ascontiguousarray(a).tobytes())
d = s.download()
for k in ("xv", "Pv", "w", "xf", "Pf"):
    h.update(np.ascontiguousarray(d[k]).tobytes())
s.close()
print("state_hash %s lib=%s" % (h.hexdigest(), os.path.basename(slam_amd.lib_path())))
