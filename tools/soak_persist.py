#!/usr/bin/env python3
"""GPU box: soak of the persistent step loop: whole example_webmap runs through slamgpu_run_observe (batches of 256 iterations, as
slam-backend hands them over) `reps` times per method in fresh contexts; every run must end without SLAMGPU_ERR_BARRIER and with
the history of the first run of its method, bit for bit.  usage: python tools/soak_persist.py [reps = 10] [N = 1000]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import slam_amd as sg  # noqa: E402
from slam_amd import host  # noqa: E402

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 10
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
f32 = np.float32
for method, mid in (("FASTSLAM1", 1), ("FASTSLAM2", 2)):
    args = ["-m", os.path.join(ROOT, "data", "example_webmap.mat"), "-method", method, "-NPARTICLES", N, "-NEFFECTIVE", int(0.75 * N), "-SWITCH_SEED_RANDOM", 7]
    tape = host.make_tape(args)
    sim = host.HostSim(args)
    lm, _ = sim.map()
    max_range = float(sim.conf.MAX_RANGE)
    sim.close()
    steps = tape["steps"]
    ctl = [np.array(st["controls"], f32).reshape(-1, 3) for st in steps]
    xt = [np.asarray(st["true"], f32) for st in steps]
    ref, iters = None, 0
    for rep in range(REPS):
        s = sg.SlamGpu(N, tape["nlm"], method=mid, n_effective=int(0.75 * N), rng_mode=sg.RNG_PHILOX, seed=5, math_mode=1, device_observe=True)
        s.set_map(lm)
        hist = []
        for a in range(0, len(steps), 256):
            b = min(len(steps), a + 256)
            s.run_observe(ctl[a:b], tape["Q"], float(tape["dt"]), xt[a:b], max_range, tape["R"], noise=2)
            if (a // 256) % 8 == 7:
                hist.append([np.asarray(x) for x in s.history_fetch()])
        hist.append([np.asarray(x) for x in s.history_fetch()])
        launches, it = s.persist_info()
        iters += it
        flat = [np.concatenate([h[k] for h in hist]) for k in range(len(hist[0]))]
        if ref is None:
            ref = flat
        else:
            for x, y in zip(ref, flat):
                assert np.array_equal(x, y, equal_nan=True), (method, rep)
        s.close()
    print("%s N=%d: %d runs of %d steps, %d iterations of the loop, no abandoned launch, histories identical" % (method, N, REPS, len(steps), iters))
