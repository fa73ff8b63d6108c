import os, re, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for exe in ("slam_amd/bin/slam-backend",):
    for N in (512, 2048):
        for seed in (7, 8, 9, 10, 11):
            for math in ("fast", "strict"):
                log = tempfile.mktemp(suffix=".csv")
                r = subprocess.run([os.path.join(ROOT, exe), "-m", os.path.join(ROOT, "data/example_webmap.mat"), "-method", "FASTSLAM2", "-NPARTICLES", str(N),
                                    "-NEFFECTIVE", str(3 * N // 4), "-SWITCH_SEED_RANDOM", str(seed), "-assoc", "gated", "-math", math, "-log", log],
                                   capture_output=True, text=True)
                nl = int(re.search(r"landmarks in map: (\d+)", r.stdout).group(1))
                rows = np.loadtxt(log, delimiter=",", skiprows=1)
                err = np.hypot(rows[:, 4] - rows[:, 1], rows[:, 5] - rows[:, 2])
                print("%-50s N=%4d seed %2d %-6s landmarks %2d  mean err %.3f  max err %.3f" % (exe, N, seed, math, nl, err.mean(), err.max()), flush=True)
