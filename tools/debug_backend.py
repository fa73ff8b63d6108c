#!/usr/bin/env python3
"""GPU box diagnostic: slam-backend -rng parity vs the same call sequence through the python binding vs the oracle."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import slam_amd as sg
from slam_amd import host
from conftest import sim_args, load_golden
g = load_golden("traj_fs2_webmap_N100_s7")
N = 100
exe = os.path.join(ROOT, "slam_amd", "bin", "slam-backend")
log = "/tmp/dbg.csv"
r = subprocess.run([exe, "-m", os.path.join(ROOT, "data", "example_webmap.mat"), "-method", "FASTSLAM2", "-rng", "parity", "-math", "strict",
                    "-NPARTICLES", "100", "-NEFFECTIVE", "75", "-SWITCH_SEED_RANDOM", "7", "-log", log, "-maxsteps", "40"], capture_output=True, text=True)
print(r.stdout[-600:], r.stderr[-300:])
rows = np.loadtxt(log, delimiter=",", skiprows=1)
for per_step_estimate in (True, False):
    h = host.HostSim(sim_args("example_webmap", "FASTSLAM2", N, 7))
    Q, R, dt = h.noise()
    s = sg.SlamGpu(N, h.nlm, method=2, n_effective=75, wheel_base=float(h.conf.WHEELBASE), sigma_phi=float(h.conf.sigmaT), rng_mode=sg.RNG_TAPE, math_mode=0)
    it, k = 0, 0
    while it < 40:
        rr, V, G, phi = h.control(); it += 1
        s.predict(V, G, Q, float(dt), phi)
        if rr == 1:
            zf, idf, zn = h.observe(s.nf())
            nm = host.draw_normals(N, 3) if (len(idf) or len(zn)) else None
            _, st = host.draw_strata(N)
            s.update(zf, idf, zn, R, nm, st)
            e = s.estimate()
            print("per_step_estimate", per_step_estimate, "obs", k, "python", e, "backend", rows[it - 1, 4:7], "golden", g["est"][k])
            k += 1
        elif per_step_estimate:
            s.estimate()
    s.close(); h.close()
