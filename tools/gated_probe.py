#!/usr/bin/env python3
"""slam-backend -assoc gated over whole example_webmap runs: seeds x builds x particle counts, with the association policy's
parameters given on the command line (host/gated.h).  A run is GOOD when it ends with at most 45 landmarks in use (the map has 35)
and a mean position error under 1 m.  usage:
    tools/gated_probe.py [--seeds 7-16] [--jobs 4] [--tag name] [-- extra slam-backend arguments, e.g. -ASSOC_NEW_SHARE 0.9]
GPU box; at most `jobs` (<= 4) runs use the card together."""
import argparse
import concurrent.futures
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "slam_amd", "bin", "slam-backend")


MAP = ["example_webmap"]


def known(N, seed, math):
    """the same run with the reference's own association (dataAssociationKnown, core.cpp:91-120): what the filter itself achieves
    with this seed's noise and this many particles"""
    log = tempfile.mktemp(suffix=".csv")
    r = subprocess.run([EXE, "-m", os.path.join(ROOT, "data", MAP[0] + ".mat"), "-method", "FASTSLAM2", "-NPARTICLES", str(N), "-NEFFECTIVE", str(3 * N // 4),
                        "-SWITCH_SEED_RANDOM", str(seed), "-math", math, "-loop", "step", "-log", log], capture_output=True, text=True)
    if r.returncode != 0:
        return 99.0
    rows = np.loadtxt(log, delimiter=",", skiprows=1)
    os.unlink(log)
    return float(np.hypot(rows[:, 4] - rows[:, 1], rows[:, 5] - rows[:, 2]).mean())


def one(N, seed, math, extra):
    log = tempfile.mktemp(suffix=".csv")
    r = subprocess.run([EXE, "-m", os.path.join(ROOT, "data", MAP[0] + ".mat"), "-method", "FASTSLAM2", "-NPARTICLES", str(N), "-NEFFECTIVE", str(3 * N // 4),
                        "-SWITCH_SEED_RANDOM", str(seed), "-assoc", "gated", "-math", math, "-log", log] + extra, capture_output=True, text=True)
    m = re.search(r"landmarks in map: (\d+)(?: \((\d+) opened, (\d+) retired)?", r.stdout)
    if r.returncode != 0 or not m:
        return (N, seed, math, -1, -1, -1, 99.0, 99.0, (r.stdout + r.stderr)[-300:], 99.0)
    rows = np.loadtxt(log, delimiter=",", skiprows=1)
    os.unlink(log)
    err = np.hypot(rows[:, 4] - rows[:, 1], rows[:, 5] - rows[:, 2])
    return (N, seed, math, int(m.group(1)), int(m.group(2) or 0), int(m.group(3) or 0), float(err.mean()), float(err.max()), "", known(N, seed, math))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="7-16")
    ap.add_argument("--jobs", type=int, default=4)
    ap.add_argument("--tag", default="")
    ap.add_argument("--particles", default="512,2048")
    ap.add_argument("--map", default="example_webmap", help="bundled map (data/<map>.mat)")
    ap.add_argument("--max-landmarks", type=int, default=45, help="GOOD: at most this many landmarks in use")
    ap.add_argument("extra", nargs="*")
    a = ap.parse_args()
    MAP[0] = a.map
    lo, hi = (int(x) for x in a.seeds.split("-"))
    jobs = [(N, seed, math) for N in (int(x) for x in a.particles.split(",")) for seed in range(lo, hi + 1) for math in ("fast", "strict")]
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(a.jobs, 4)) as ex:
        res = list(ex.map(lambda j: one(*j, a.extra), jobs))
    good = rel = 0
    for (N, seed, math, nl, opened, retired, em, ex_, msg, ek) in res:
        ok = 0 <= nl <= a.max_landmarks and em < 1.0
        ok_rel = 0 <= nl <= a.max_landmarks and em <= 1.2 * ek + 0.05
        good += ok
        rel += ok_rel
        print("%-28s N=%4d seed %2d %-6s landmarks in use %2d (opened %2d, retired %2d)  mean err %.3f  max err %.3f  known-association mean err %.3f  %s %s %s"
              % (a.tag or " ".join(a.extra), N, seed, math, nl, opened, retired, em, ex_, ek, "good" if ok else "BAD", "like-known" if ok_rel else "WORSE-THAN-KNOWN", msg), flush=True)
    print("%-28s GOOD %d of %d (landmarks in use within --max-landmarks and mean position error < 1 m); %d of %d within 1.2 x + 0.05 m of the SAME run with the reference's "
          "known association; mean of the mean errors %.3f m (known association: %.3f m), median landmarks %d"
          % (a.tag or " ".join(a.extra), good, len(res), rel, len(res), float(np.mean([r[6] for r in res])), float(np.mean([r[9] for r in res])),
             int(np.median([r[3] for r in res]))), flush=True)


if __name__ == "__main__":
    sys.exit(main())
