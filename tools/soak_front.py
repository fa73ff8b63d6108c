#!/usr/bin/env python3
"""GPU box soak: the in-launch observation front end (slamgpu_step_observe on a compact context) against the same run stepped
with slamgpu_step on the packets it made, over seeds, particle counts (also odd ones), methods and builds; every state array
and history must be bit-identical.  Reads in the middle of the device-driven run move the bookkeeping to the host and back.

usage: python tools/soak_front.py [cases]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import slam_amd as sg  # noqa: E402
from slam_amd import host  # noqa: E402

CASES = int(sys.argv[1]) if len(sys.argv) > 1 else 24
f32 = np.float32
rng = np.random.default_rng(2024)
maps = ["example_webmap", "example_loop1", "example_loop2"]
bad = 0
for case in range(CASES):
    mapname = maps[case % len(maps)]
    method = 2 if rng.random() < 0.7 else 1
    N = int(rng.choice([256, 1000, 2048, 4999, 10000, 33333]))
    seed = int(rng.integers(1, 1000))
    math_mode = int(rng.integers(0, 2))
    nobs = int(rng.integers(150, 900))
    args = ["-m", os.path.join(ROOT, "data", mapname + ".mat"), "-method", "FASTSLAM2" if method == 2 else "FASTSLAM1", "-NPARTICLES", 100,
            "-NEFFECTIVE", 75, "-SWITCH_SEED_RANDOM", seed]
    tape = host.make_tape(args, max_obs=nobs)
    sim = host.HostSim(args)
    lm, _ = sim.map()
    max_range = float(sim.conf.MAX_RANGE)
    wb, sp = float(sim.conf.WHEELBASE), float(sim.conf.sigmaT)
    sim.close()
    Q, R, dt = tape["Q"], tape["R"], float(tape["dt"])
    kw = dict(method=method, n_effective=int(0.75 * N), rng_mode=sg.RNG_PHILOX, seed=seed, math_mode=math_mode, wheel_base=wb, sigma_phi=sp)
    a = sg.SlamGpu(N, tape["nlm"], device_observe=True, **kw)
    a.set_map(lm)
    packets = []
    pk = int(rng.integers(20, 120))
    for i, st in enumerate(tape["steps"]):
        a.step_observe(np.array(st["controls"], f32).reshape(-1, 3), Q, dt, st["true"], max_range, R, noise=2)
        packets.append(a.observe_fetch())
        if i % pk == pk - 1:
            what = rng.random()
            if what < 0.4:
                a.peek(first=int(rng.integers(0, 7)), stride=int(rng.integers(3, 50)))
            elif what < 0.7:
                a.nf()
            elif what < 0.85:
                a.download()        # (flattens the genealogy: every landmark back in row 0, the book pushed again)
            else:
                a.estimate()
    ha, rows, da = a.history_fetch(), a.live_rows(), a.download()
    a.close()
    b = sg.SlamGpu(N, tape["nlm"], **kw)
    for st, p in zip(tape["steps"], packets):
        b.step(np.array(st["controls"], f32).reshape(-1, 3), Q, dt, p["zf"], p["idf"], p["zn"], R)
    hb, db = b.history_fetch(), b.download()
    b.close()
    ok = da["nf"] == db["nf"] and all(np.array_equal(x, y, equal_nan=True) for x, y in zip(ha, hb))
    for key in ("xv", "Pv", "w", "xf", "Pf"):
        ok = ok and np.array_equal(da[key].view(np.uint32), db[key].view(np.uint32))
    print("case %2d %-14s method %d N %5d seed %3d math %d steps %3d nf %2d rows %2d resamples %3d max m %2d: %s"
          % (case, mapname, method, N, seed, math_mode, len(tape["steps"]), da["nf"], rows, int(ha[2].sum()), max(p["zf"].shape[0] for p in packets),
             "ok" if ok else "MISMATCH"), flush=True)
    bad += 0 if ok else 1
print("%d of %d cases differ" % (bad, CASES))
sys.exit(1 if bad else 0)
