#!/usr/bin/env python3
"""BASELINE config 4 on one GPU (1 001 472 particles, example_webmap, fast build): the whole run in chunks of 100 observation
steps -- wall time per step, the update launch's and scan_kernel's own durations (HIP event pairs: a second, profiled pass), the
step's content (m, births, landmarks, resamples), genealogy rows in use and the shader clock -- to find what bench.py's window
(80.7 us per step in round 5) does not see of the run (88.3): VERDICT r5 item 6.  GPU box:
    python3 tools/c4_whole_run.py [particles] > profiles/config4_whole_run_r06.txt"""
import glob
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import slam_amd  # noqa: E402
from slam_amd import host  # noqa: E402


def sclk():
    """current shader clock (MHz) of the first card that says: the starred level of pp_dpm_sclk"""
    for f in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")):
        try:
            for ln in open(f):
                if "*" in ln:
                    return int(re.search(r"(\d+)\s*Mhz", ln, re.I).group(1))
        except (OSError, AttributeError):
            pass
    return 0


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1001472
    chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    tape = host.make_tape(["-m", os.path.join(ROOT, "data", "example_webmap.mat"), "-method", "FASTSLAM2", "-NPARTICLES", N, "-NEFFECTIVE", int(0.75 * N),
                           "-SWITCH_SEED_RANDOM", 7])
    obs = tape["steps"]
    Q, R, dt = tape["Q"], tape["R"], float(tape["dt"])
    rows = {}
    for profiled in (False, True):
        s = slam_amd.SlamGpu(N, tape["nlm"], method=slam_amd.FASTSLAM2, n_effective=int(0.75 * N), wheel_base=float(tape["conf"].WHEELBASE),
                             sigma_phi=float(tape["conf"].sigmaT), rng_mode=slam_amd.RNG_PHILOX, seed=7, math_mode=slam_amd.MATH_FAST)
        calls = [s.prepare_step(np.array(st["controls"], np.float32).reshape(-1, 3), Q, dt, st["zf"], st["idf"], st["zn"], R) for st in obs]
        if profiled:
            s.profile(True)
        prev = {"fs2_update": (0.0, 0), "scan": (0.0, 0)}
        for c0 in range(0, len(obs), chunk):
            c1 = min(len(obs), c0 + chunk)
            s.sync()
            clk0 = sclk()
            t0 = time.perf_counter()
            for k in range(c0, c1):
                calls[k]()
                if profiled and (k & 31) == 31:
                    s.sync()
            s.sync()
            wall = (time.perf_counter() - t0) / (c1 - c0)
            est, neff, res = s.history_fetch()
            r = rows.setdefault(c0, {})
            if not profiled:
                r.update(wall_us=1e6 * wall, m=float(np.mean([obs[k]["zf"].shape[0] for k in range(c0, c1)])), n=float(np.sum([obs[k]["zn"].shape[0] for k in range(c0, c1)])),
                         nf=int(obs[c1 - 1]["nf_before"] + obs[c1 - 1]["zn"].shape[0]), res=float(np.mean(res)), rows=s.live_rows(), sclk=(clk0, sclk()))
            else:
                for name in ("fs2_update", "scan"):
                    ms, cnt = s.kernel_time(name)
                    pm, pc = prev[name]
                    r[name] = 1e3 * (ms - pm) / max(cnt - pc, 1)
                    prev[name] = (ms, cnt)
        s.close()
    print("config 4 on one GPU: %d particles, example_webmap, fast build, Philox; chunks of %d observation steps" % (N, chunk))
    print("%9s %8s %8s %8s %6s %6s %4s %6s %5s %s" % ("steps", "wall us", "update", "scan", "m", "births", "nf", "resamp", "rows", "sclk MHz (before, after)"))
    for c0 in sorted(rows):
        r = rows[c0]
        print("%4d-%4d %8.2f %8.2f %8.2f %6.2f %6.0f %4d %6.2f %5d %s" % (c0, min(len(obs), c0 + chunk), r["wall_us"], r.get("fs2_update", 0.0), r.get("scan", 0.0), r["m"], r["n"],
                                                                      r["nf"], r["res"], r["rows"], r["sclk"]))
    w = np.array([rows[c]["wall_us"] for c in sorted(rows)])
    print("whole run: mean of the chunks %.2f us per step; chunk at step 1000: %.2f; min chunk %.2f, max chunk %.2f" % (w.mean(), rows[1000 // chunk * chunk]["wall_us"], w.min(), w.max()))
    # what explains a chunk's time: least squares on m, resample rate and landmarks in the map
    A = np.array([[1.0, rows[c]["m"], rows[c]["res"], rows[c]["nf"]] for c in sorted(rows)])
    coef, *_ = np.linalg.lstsq(A, w, rcond=None)
    print("least squares over the chunks: wall us per step = %.2f + %.2f m + %.2f resample rate + %.3f landmarks in the map (residual rms %.2f us)"
          % (coef[0], coef[1], coef[2], coef[3], float(np.sqrt(np.mean((A @ coef - w) ** 2)))))


if __name__ == "__main__":
    main()
