#!/usr/bin/env python3
"""GPU box diagnostic: k queued predicts applied by one launch vs one stand-alone launch each (sync / estimate between)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import slam_amd as sg
from slam_amd import host
from conftest import sim_args
N = 100
h = host.HostSim(sim_args("example_webmap", "FASTSLAM2", N, 7))
Q, R, dt = h.noise()
ctl = []
for _ in range(8):
    r, V, G, phi = h.control(); ctl.append((V, G, phi))
print("controls", ctl)
def run(mode, mm):
    s = sg.SlamGpu(N, 35, method=2, n_effective=75, wheel_base=float(h.conf.WHEELBASE), sigma_phi=float(h.conf.sigmaT), rng_mode=sg.RNG_TAPE, math_mode=mm)
    outs = []
    for k, (V, G, phi) in enumerate(ctl):
        s.predict(V, G, Q, float(dt), phi)
        if mode == "sync":
            s.sync()
        elif mode == "estimate":
            outs.append(s.estimate())
        elif mode == "download":
            outs.append(s.download(landmarks=False)["xv"][0].copy())
    d = s.download(landmarks=False)
    s.close()
    return d, outs
for mm in (0, 1):
    ref, _ = run("fused", mm)
    for mode in ("sync", "estimate", "download", "fused"):
        d, outs = run(mode, mm)
        print("math", mm, mode, "xv[0]", d["xv"][0], "max|dxv|", np.abs(d["xv"] - ref["xv"]).max(), "max|dPv|", np.abs(d["Pv"] - ref["Pv"]).max(),
              "all particles equal:", bool(np.all(d["xv"] == d["xv"][0])))
        if outs:
            print("   per-step:", [np.round(o, 5).tolist() for o in outs])
