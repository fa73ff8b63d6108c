#!/usr/bin/env python3
"""GPU box soak: the whole example_webmap run (2 172 observation steps) through distributed contexts -- logical shards
with the gather collective (G = 4) and with the push collective (G = 2) -- against one context: estimates, Neff, decisions
of every step and the final particle set must be identical.  usage: python tools/dist_soak.py [N]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import slam_amd as sg  # noqa: E402
from slam_amd import host  # noqa: E402
from slam_amd.dist import DistFilter  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
tp = host.make_tape(["-m", os.path.join(ROOT, "data", "example_webmap.mat"), "-method", "FASTSLAM2", "-NPARTICLES", N, "-NEFFECTIVE", int(0.75 * N),
                     "-SWITCH_SEED_RANDOM", 7])
kw = dict(method=sg.FASTSLAM2, n_effective=int(0.75 * N), seed=7, math_mode=1)


def drive(f):
    hs = []
    for k, st in enumerate(tp["steps"]):
        f.step(np.array(st["controls"], np.float32).reshape(-1, 3), tp["Q"], float(tp["dt"]), st["zf"], st["idf"], st["zn"], tp["R"])
        if (k & 1023) == 1023:
            hs.append(f.history_fetch())
    hs.append(f.history_fetch())
    return [np.concatenate([h[j] for h in hs]) for j in range(3)]


s = sg.SlamGpu(N, tp["nlm"], rng_mode=sg.RNG_PHILOX, **kw)
href = drive(s)
ref = s.download()
s.close()
print("single context: %d steps, %d resamples" % (len(href[0]), int(href[2].sum())))
for G, push in ((4, False), (2, True)):
    f = DistFilter.local(G, N // G, tp["nlm"], **kw)
    if push:
        assert f.use_push()
    h = drive(f)
    parts = f.download()
    if push:
        assert f.collective_ok()
    f.close()
    assert np.array_equal(h[1], href[1]) and np.array_equal(h[2], href[2]), "Neff / decision history differs"
    assert np.abs(h[0] - href[0]).max() <= 1e-11, np.abs(h[0] - href[0]).max()
    for key in ("xv", "Pv", "w", "xf", "Pf"):
        cat = np.concatenate([p[key] for p in parts])
        assert np.array_equal(cat.view(np.uint32), ref[key].view(np.uint32)), key
    print("G=%d %s: identical over %d steps (max estimate difference %.2e)" % (G, "push" if push else "gather", len(h[0]), np.abs(h[0] - href[0]).max()))
print("DIST_SOAK_OK")
