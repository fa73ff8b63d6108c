#!/usr/bin/env python3
"""GPU box diagnostic: is the step loop deterministic?  The first steps of the example_webmap run at N particles, driven like
tests/test_gpu_parity.py::test_full_size_philox_vs_oracle (download + stats + ancestors after every update: the resampling stage
runs as its own launch), REPS times in fresh contexts; every downloaded array is compared bit for bit with the first repetition.

usage: python tools/determinism.py [N] [reps] [steps] [math_mode]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import slam_amd as sg  # noqa: E402
from slam_amd import host  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 20
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 8
MODE = int(sys.argv[4]) if len(sys.argv) > 4 else 0
f32 = np.float32
tape = host.make_tape(["-m", os.path.join(ROOT, "data", "example_webmap.mat"), "-method", "FASTSLAM2", "-NPARTICLES", N,
                       "-NEFFECTIVE", int(0.75 * N), "-SWITCH_SEED_RANDOM", 7], max_obs=STEPS)
Q, R, dt = tape["Q"], tape["R"], float(tape["dt"])
ref = None
bad = 0
for rep in range(REPS):
    s = sg.SlamGpu(N, tape["nlm"], method=2, n_effective=int(0.75 * N), rng_mode=sg.RNG_PHILOX, seed=7, math_mode=MODE)
    outs = []
    for k, st in enumerate(tape["steps"]):
        for (V, G, phi) in np.array(st["controls"], f32).reshape(-1, 3):
            s.predict(float(V), float(G), Q, dt, float(phi))
        s.update(st["zf"], st["idf"], st["zn"], R)
        d = s.download()
        ne, did, wsum = s.stats()
        keep = s.ancestors() if did else np.zeros(0, np.int32)
        outs.append(dict(xv=d["xv"].copy(), w=d["w"].copy(), xf=d["xf"].copy(), Pf=d["Pf"].copy(), keep=keep.copy(), ne=np.array([ne, did, wsum])))
    s.close()
    if ref is None:
        ref = outs
        print("reference repetition: resampled at steps", [k for k, o in enumerate(outs) if o["ne"][1]])
        continue
    for k, (a, b) in enumerate(zip(ref, outs)):
        for key in a:
            x, y = a[key], b[key]
            same = np.array_equal(x.view(np.uint32) if x.dtype == np.float32 else x, y.view(np.uint32) if y.dtype == np.float32 else y)
            if not same:
                bad += 1
                diff = (x != y)
                idx = np.argwhere(diff.reshape(diff.shape[0], -1).any(axis=1)).ravel()
                print("rep %d step %d %s: %d rows differ; first %s last %s" % (rep, k, key, idx.size, idx[:8], idx[-4:]))
                break
        else:
            continue
        break
print("%d of %d repetitions differ from the first" % (bad, REPS - 1))
