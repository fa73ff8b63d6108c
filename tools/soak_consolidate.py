#!/usr/bin/env python3
"""GPU box soak: row consolidation on against off (SLAMGPU_NO_CONSOLIDATE), over maps, seeds, particle counts, methods, builds and
thresholds: histories, final state and a mid-run view must be bit-identical.

usage: python tools/soak_consolidate.py [cases]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import slam_amd as sg  # noqa: E402
from slam_amd import host  # noqa: E402

CASES = int(sys.argv[1]) if len(sys.argv) > 1 else 18
f32 = np.float32
rng = np.random.default_rng(77)
maps = ["example_webmap", "example_loop1", "example_loop902", "example_loop2"]
bad = 0
for case in range(CASES):
    mapname = maps[case % len(maps)]
    method = 2 if rng.random() < 0.7 else 1
    N = int(rng.choice([256, 1000, 2048, 4999, 20000]))
    seed = int(rng.integers(1, 1000))
    math_mode = int(rng.integers(0, 2))
    nobs = int(rng.integers(200, 1500))
    args = ["-m", os.path.join(ROOT, "data", mapname + ".mat"), "-method", "FASTSLAM2" if method == 2 else "FASTSLAM1", "-NPARTICLES", 100,
            "-NEFFECTIVE", 75, "-SWITCH_SEED_RANDOM", seed]
    tape = host.make_tape(args, max_obs=nobs)
    sim = host.HostSim(args)
    wb, sp = float(sim.conf.WHEELBASE), float(sim.conf.sigmaT)
    sim.close()
    Q, R, dt = tape["Q"], tape["R"], float(tape["dt"])
    kw = dict(method=method, n_effective=int(0.75 * N), rng_mode=sg.RNG_PHILOX, seed=seed, math_mode=math_mode, wheel_base=wb, sigma_phi=sp)
    above, target = int(rng.integers(1, 9)), int(rng.integers(2, 30))
    out = []
    peek_at = int(rng.integers(50, len(tape["steps"]) - 10))
    for off in (False, True):
        os.environ["SLAMGPU_CONSOLIDATE_ABOVE"] = str(above)
        os.environ["SLAMGPU_PLAIN_ROWS_TARGET"] = str(target)
        if off:
            os.environ["SLAMGPU_NO_CONSOLIDATE"] = "1"
        else:
            os.environ.pop("SLAMGPU_NO_CONSOLIDATE", None)
        s = sg.SlamGpu(N, tape["nlm"], **kw)
        mid = None
        for i, st in enumerate(tape["steps"]):
            s.step(np.array(st["controls"], f32).reshape(-1, 3), Q, dt, st["zf"], st["idf"], st["zn"], R)
            if i == peek_at:
                mid = s.peek(first=1, stride=5)
        h, rows = s.history_fetch(), s.live_rows()
        out.append((s.download(), h, mid, rows))
        s.close()
    os.environ.pop("SLAMGPU_NO_CONSOLIDATE", None)
    (a, ha, ma, ra), (b, hb, mb, rb) = out
    ok = a["nf"] == b["nf"] and all(np.array_equal(x, y, equal_nan=True) for x, y in zip(ha, hb))
    for key in ("xv", "Pv", "w", "xf", "Pf"):
        ok = ok and np.array_equal(a[key].view(np.uint32), b[key].view(np.uint32)) and np.array_equal(ma[key].view(np.uint32), mb[key].view(np.uint32))
    print("case %2d %-15s method %d N %5d seed %3d math %d steps %4d nf %3d rows %3d (off: %3d) above %d target %2d resamples %4d: %s"
          % (case, mapname, method, N, seed, math_mode, len(tape["steps"]), a["nf"], ra, rb, above, target, int(ha[2].sum()), "ok" if ok else "MISMATCH"), flush=True)
    bad += 0 if ok else 1
print("%d of %d cases differ" % (bad, CASES))
sys.exit(1 if bad else 0)
