#!/usr/bin/env python3
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import slam_amd as sg
from slam_amd import host
from conftest import sim_args, load_golden
g = load_golden("traj_fs2_webmap_N100_s7")
N = 100
def flow(mode):
    h = host.HostSim(sim_args("example_webmap", "FASTSLAM2", N, 7))
    Q, R, dt = h.noise()
    s = sg.SlamGpu(N, h.nlm, method=2, n_effective=75, wheel_base=float(h.conf.WHEELBASE), sigma_phi=float(h.conf.sigmaT), rng_mode=sg.RNG_TAPE, math_mode=0)
    it, k, out = 0, 0, []
    while k < 2:
        rr, V, G, phi = h.control(); it += 1
        s.predict(V, G, Q, float(dt), phi)
        if rr == 1:
            zf, idf, zn = h.observe(s.nf())
            nm = host.draw_normals(N, 3) if (len(idf) or len(zn)) else None
            _, st = host.draw_strata(N)
            if mode == "pre-download":
                pre = s.download(landmarks=False)
            s.update(zf, idf, zn, R, nm, st)
            e = s.estimate()
            d = s.download()
            out.append((e, d, nm))
            k += 1
        elif mode == "estimate":
            s.estimate()
        elif mode == "sync":
            s.sync()
    s.close(); h.close()
    return out
A = flow("none")
for mode in ("estimate", "sync", "pre-download"):
    Bf = flow(mode)
    for k in range(2):
        (ea, da, na), (eb, db, nb) = A[k], Bf[k]
        print(mode, "obs", k, "est A", ea, "est B", eb, "golden", g["est"][k], "normals equal", np.array_equal(na, nb))
        for key in ("xv", "Pv", "w", "xf"):
            print("    ", key, "max|diff|", np.abs(da[key] - db[key]).max(), " A[0]", da[key][0].ravel()[:4], " B[0]", db[key][0].ravel()[:4])
