#!/usr/bin/env python3
"""GPU box diagnostic (round 5): where an ITERATION of the persistent step loop spends its time (slamgpu_run_observe on a small
compact context: one launch for K iterations, kernels.h: PersistArgs).  Instrumented library as tools/stamps.py; the stamps a
launch leaves are those of its LAST iteration, so every sample is one call of K iterations ending at another step of the run.
Levels 0..10 as in tools/stamps.py (0 = the iteration's code begins), 12 = iteration begins (previous barrier passed), 11 = the
step's code is done (stores issued), 13 = barrier passed (stores drained, everybody arrived, L1 invalidated).

usage: python tools/stamps_persist.py [N] [samples] [FASTSLAM1|FASTSLAM2] [K] [flow]
       flow: stamps that do not drain (`make -C slam_amd/csrc stamps_flow`): the iteration's own schedule"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FLOW = len(sys.argv) > 5 and sys.argv[5] == "flow"
os.environ["SLAMGPU_LIB"] = os.path.join(ROOT, "slam_amd", "libslamgpu_stamps_flow.so" if FLOW else "libslamgpu_stamps.so")
os.environ["SLAMGPU_STAMPS"] = "1"
import numpy as np  # noqa: E402
import slam_amd  # noqa: E402
from slam_amd import host  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
SAMPLES = int(sys.argv[2]) if len(sys.argv) > 2 else 100
METHOD = sys.argv[3] if len(sys.argv) > 3 else "FASTSLAM1"
K = int(sys.argv[4]) if len(sys.argv) > 4 else 6
MID = 2 if METHOD == "FASTSLAM2" else 1
START = 1000
NAMES = {0: "0 step code begins", 1: "1 Ctrl words / head loads requested", 2: "2 block totals scanned (W, Neff, decision)", 3: "3 ancestor found",
         4: "4 pose + genealogy arrived", 10: "10 queued predicts applied", 5: "5 records staged in LDS", 6: "6 proposal pass done",
         7: "7 landmark pass done, record stores landed", 8: "8 pose / genealogy stores landed", 9: "9 weight prefix + totals written",
         11: "11 step code done", 13: "13 barrier passed (end of the iteration)"}
ORDER = [12, 0, 1, 2, 3, 4, 10, 5, 6, 7, 8, 9, 11, 13]
NAMES[12] = "12 iteration begins (= 0)"

args = ["-m", os.path.join(ROOT, "data", "example_webmap.mat"), "-method", METHOD, "-NPARTICLES", N, "-NEFFECTIVE", int(0.75 * N), "-SWITCH_SEED_RANDOM", 7]
tape = host.make_tape(args, max_obs=START + SAMPLES * K + 2)
sim = host.HostSim(args)
LM, _ = sim.map()
MAX_RANGE = float(sim.conf.MAX_RANGE)
sim.close()
Q, R, dt = tape["Q"], tape["R"], float(tape["dt"])
steps = tape["steps"]
ctl = [np.array(st["controls"], np.float32).reshape(-1, 3) for st in steps]
xt = [np.asarray(st["true"], np.float32) for st in steps]
s = slam_amd.SlamGpu(N, tape["nlm"], method=MID, n_effective=int(0.75 * N), rng_mode=slam_amd.RNG_PHILOX, seed=7, math_mode=1, device_observe=True)
s.set_map(LM)
for a in range(0, START, 500):
    s.run_observe(ctl[a:a + 500], Q, dt, xt[a:a + 500], MAX_RANGE, R, noise=2)
    s.estimate_fetch()
sets = []
k = START
for _ in range(SAMPLES):
    s.run_observe(ctl[k:k + K], Q, dt, xt[k:k + K], MAX_RANGE, R, noise=2)
    st = s.debug_stamps().astype(np.int64)
    _, _, res = s.history_fetch()
    sets.append((st, bool(res[-2]) if len(res) >= 2 else False))   # did the step BEFORE the last one resample (the last iteration applies it)
    k += K
print("persistent step loop, %s, N = %d (%d workgroups + helper), K = %d iterations per launch, steps %d.., fast build, instrumented" % (METHOD, N, (N + 255) // 256, K, START))
print("launches / iterations of the loop:", s.persist_info(cross=True))
s.close()
for title, sel in (("last iteration applied a RESAMPLE", True), ("last iteration only normalised", False)):
    grp = [st for st, r in sets if r == sel]
    print("\n== %s (%d launches); microseconds since the first workgroup began the iteration" % (title, len(grp)))
    if not grp:
        continue
    rel = np.concatenate([(st - st[:, 12].min()) / 100.0 for st in grp])
    prev = None
    print("%-52s %8s %8s %8s   %s" % ("level", "median", "p90", "max", "median step from previous level"))
    for j in ORDER:
        col = rel[:, j]
        med = np.median(col)
        if med < -1 or med > 1e6:
            continue
        print("%-52s %8.2f %8.2f %8.2f   %s" % (NAMES[j], med, np.quantile(col, 0.9), col.max(), "" if prev is None else "%+.2f" % (med - prev)))
        prev = med
    # per workgroup: the medians of a few levels (the barrier waits for the slowest)
    per = np.stack([(st - st[:, 12].min()) / 100.0 for st in grp])   # [launch][workgroup][slot]
    print("per workgroup, medians:  " + "  ".join("%s" % NAMES[j].split()[0] for j in (12, 2, 3, 4, 10, 7, 8, 9, 11, 13)))
    for b in range(per.shape[1]):
        print("  workgroup %d:           " % b + "  ".join("%5.2f" % np.median(per[:, b, j]) for j in (12, 2, 3, 4, 10, 7, 8, 9, 11, 13)))
