#!/usr/bin/env python3
"""GPU box check at bench scale: G logical shards of 100 096 particles on ONE GPU (LocalComm: device-to-device copies
stand in for the collectives) against a single context holding all of them: pose estimates and final state bit-identical."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import slam_amd as sg
from slam_amd import host
from slam_amd.sharded import GpuEngine, LocalComm, ShardedFilter
G = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = 100096
N = n * G
nobs = int(sys.argv[2]) if len(sys.argv) > 2 else 400
tape = host.make_tape(["-m", os.path.join(ROOT, "data", "example_webmap.mat"), "-method", "FASTSLAM2", "-NPARTICLES", N, "-NEFFECTIVE", int(0.75 * N), "-SWITCH_SEED_RANDOM", 7], max_obs=nobs)
Q, R, dt = tape["Q"], tape["R"], float(tape["dt"])
ctl = [np.array(st["controls"], np.float32).reshape(-1, 3) for st in tape["steps"]]
s = sg.SlamGpu(N, tape["nlm"], method=2, n_effective=int(0.75 * N), rng_mode=sg.RNG_PHILOX, seed=7, math_mode=1)
for k, st in enumerate(tape["steps"]):
    s.step(ctl[k], Q, dt, st["zf"], st["idf"], st["zn"], R)
e1, _, r1 = s.history_fetch()
ref = s.download()
s.close()
eng = [GpuEngine(g, G, n, tape["nlm"], method=2, n_effective=int(0.75 * N), rng_mode=sg.RNG_PHILOX, seed=7, math_mode=1) for g in range(G)]
flt = ShardedFilter(eng, LocalComm(eng), G)
res = []
t0 = time.perf_counter()
for k, st in enumerate(tape["steps"]):
    res.append(bool(flt.step(ctl[k], Q, dt, st["zf"], st["idf"], st["zn"], R).resampled))
est = flt.estimate_fetch()
dt_ = time.perf_counter() - t0
ds = [e.ctx.download() for e in eng]
moved = flt.exchanged_records
flt.close()
ok = res == [bool(x) for x in r1] and np.abs(est - e1).max() < 1e-9
for key in ("xv", "Pv", "w", "xf", "Pf"):
    got = np.concatenate([d[key] for d in ds])
    ok = ok and np.array_equal(got.view(np.uint32), ref[key].view(np.uint32))
print("G=%d x %d particles, %d steps, %d resamples, %d offspring crossed shard boundaries, max |estimate diff| %.2e, state bit-identical: %s (%.0f us/step through LocalComm)"
      % (G, n, nobs, sum(res), moved, np.abs(est - e1).max(), ok, 1e6 * dt_ / nobs))
sys.exit(0 if ok else 1)
