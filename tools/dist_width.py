#!/usr/bin/env python3
"""GPU box: what the width of the gathered totals table costs a distributed launch.  G logical shards of 100 096 particles
on ONE GPU and one stream run their launches back to back, so (time per step) / G is the launch time of one shard when
the particle set spans G GPUs (remote reads excepted: everything is local here).
usage: python tools/dist_width.py [particles per shard] [G,G,...]   (under rocprofv3 --kernel-trace --stats with ONE G: the average
duration of update_kernel<2, 2, false> is the cleaner number: the wall time includes G host loops)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import slam_amd as sg  # noqa: E402
from slam_amd import host  # noqa: E402

n, start, steps = (int(sys.argv[1]) if len(sys.argv) > 1 else 100096), 600, 400
for G in ([int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else (1, 2, 4, 8)):
    Np = G * n
    tp = host.make_tape(["-m", os.path.join(ROOT, "data", "example_webmap.mat"), "-method", "FASTSLAM2", "-NPARTICLES", Np, "-NEFFECTIVE",
                         int(0.75 * Np), "-SWITCH_SEED_RANDOM", 7], max_obs=start + steps)
    g = sg.DistGroup(G, n, tp["nlm"], method=sg.FASTSLAM2, n_effective=int(0.75 * Np), seed=7, math_mode=1)
    calls = [g.prepare_step(np.array(st["controls"], np.float32).reshape(-1, 3), tp["Q"], float(tp["dt"]), st["zf"], st["idf"], st["zn"], tp["R"])
             for st in tp["steps"]]
    for c in calls[:start]:
        c()
    g.history_fetch()
    g.sync()
    t0 = time.perf_counter()
    for c in calls[start:start + steps]:
        c()
    g.sync()
    dt = time.perf_counter() - t0
    _, _, res = g.history_fetch()
    print("G=%d: %.2f us per step, %.2f us per shard launch (+ 1/G of the gather kernel); resample rate %.2f" % (G, 1e6 * dt / steps, 1e6 * dt / steps / G, res.mean()))
    g.close()
