#!/usr/bin/env python3
"""GPU box diagnostic: does the HIP runtime consume libc rand()?  (The oracle and the host front end draw the reference's random
numbers from libc rand(), as the reference does; anything else calling rand() in between shifts their stream.)"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
libc = ctypes.CDLL("libc.so.6")
libc.srand(12345)
expect = [libc.rand() for _ in range(200000)]   # the stream with nothing in between
libc.srand(12345)
pos = 0


def probe(tag):
    global pos
    v = libc.rand()
    try:
        at = expect.index(v, pos)
    except ValueError:
        at = -1
    print("%-44s rand() is at position %d (expected %d): %s" % (tag, at, pos, "ok" if at == pos else "SHIFTED by %d" % (at - pos)), flush=True)
    if at >= 0:
        pos = at + 1
    else:           # the generator was re-seeded by somebody: start over
        libc.srand(12345)
        pos = 0


probe("start")
import numpy as np  # noqa: E402
probe("after numpy import")
import slam_amd as sg  # noqa: E402
probe("after slam_amd import")
n = sg.device_count()
probe("after device_count")
from slam_amd import host  # noqa: E402
tape = host.make_tape(["-m", os.path.join(ROOT, "data", "example_webmap.mat"), "-method", "FASTSLAM2", "-NPARTICLES", 4096,
                       "-NEFFECTIVE", 3072, "-SWITCH_SEED_RANDOM", 7], max_obs=40)
libc.srand(12345)
pos = 0
probe("after make_tape (re-seeded)")
s = sg.SlamGpu(100000, tape["nlm"], method=2, n_effective=75000, rng_mode=sg.RNG_PHILOX, seed=7, math_mode=0)
probe("after context creation")
f32 = np.float32
for k, st in enumerate(tape["steps"][:30]):
    s.step(np.array(st["controls"], f32).reshape(-1, 3), tape["Q"], float(tape["dt"]), st["zf"], st["idf"], st["zn"], tape["R"])
    if k < 6 or k % 8 == 0:
        probe("after step %d" % k)
    if k == 10:
        s.download()
        probe("after download")
        s.stats()
        probe("after stats")
s.sync()
probe("after sync")
s.close()
probe("after close")
