#!/usr/bin/env python3
"""Does a 20-step window of BASELINE config 4 on one GPU (10^6 particles, ~80 us per step: 1.6 ms of work) run at another clock
than the whole run (190 ms of the same work)?  One context, advanced to the window; then the SAME kind of region timed as bursts of
20, 50, 100, 200, 500 and 1 000 consecutive steps, each after 100 ms of idle, while a thread samples the card's shader clock and
power from hwmon.  Content differs a little between regions (a step costs 23 + 10.9 m + 46 r us: profiles/config4_whole_run_r06.txt), so
every region also reports what that model predicts for its steps.  GPU box:  python3 tools/c4_clock_probe.py"""
import glob
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import slam_amd  # noqa: E402
from slam_amd import host  # noqa: E402


def hw(name):
    for f in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/" + name)):
        return f
    return None


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.f, self.p = hw("freq1_input"), hw("power1_input") or hw("power1_average")
        self.rows, self.on = [], True

    def run(self):
        while self.on:
            try:
                fr = int(open(self.f).read()) / 1e6 if self.f else 0.0
                pw = int(open(self.p).read()) / 1e6 if self.p else 0.0
            except (OSError, ValueError):
                fr = pw = 0.0
            self.rows.append((time.perf_counter(), fr, pw))
            time.sleep(0.001)

    def between(self, t0, t1):
        sel = [(f, p) for (t, f, p) in self.rows if t0 <= t <= t1]
        if not sel:
            return 0.0, 0.0, 0
        return float(np.mean([s[0] for s in sel])), float(np.mean([s[1] for s in sel])), len(sel)


def main():
    N = 1001472
    tape = host.make_tape(["-m", os.path.join(ROOT, "data", "example_webmap.mat"), "-method", "FASTSLAM2", "-NPARTICLES", N, "-NEFFECTIVE", int(0.75 * N),
                           "-SWITCH_SEED_RANDOM", 7])
    obs = tape["steps"]
    Q, R, dt = tape["Q"], tape["R"], float(tape["dt"])
    s = slam_amd.SlamGpu(N, tape["nlm"], method=slam_amd.FASTSLAM2, n_effective=int(0.75 * N), wheel_base=float(tape["conf"].WHEELBASE),
                         sigma_phi=float(tape["conf"].sigmaT), rng_mode=slam_amd.RNG_PHILOX, seed=7, math_mode=slam_amd.MATH_FAST)
    calls = [s.prepare_step(np.array(st["controls"], np.float32).reshape(-1, 3), Q, dt, st["zf"], st["idf"], st["zn"], R) for st in obs]
    smp = Sampler()
    smp.start()
    print("hwmon: clock %s, power %s" % (smp.f, smp.p))
    k = 0
    for _ in range(100):
        calls[k]()
        k += 1
    s.sync()
    s.history_fetch()
    print("%8s %10s %12s %10s %10s %8s" % ("steps", "us / step", "model us", "sclk MHz", "power W", "samples"))
    for L in (20, 20, 50, 100, 200, 500, 1000, 20, 20, 20):
        if k + L > len(calls):
            break
        time.sleep(0.1)
        s.sync()
        t0 = time.perf_counter()
        for j in range(k, k + L):
            calls[j]()
        s.sync()
        t1 = time.perf_counter()
        m = np.mean([obs[j]["zf"].shape[0] for j in range(k, k + L)])
        _, _, res = s.history_fetch()
        model = 23.47 + 10.92 * m + 45.98 * float(np.mean(res)) + 0.115 * 35
        f, p, ns = smp.between(t0, t1)
        print("%8d %10.2f %12.2f %10.0f %10.0f %8d" % (L, 1e6 * (t1 - t0) / L, model, f, p, ns), flush=True)
        k += L
    smp.on = False
    s.close()


if __name__ == "__main__":
    main()
