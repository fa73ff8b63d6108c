#!/usr/bin/env python3
"""GPU box diagnostic: where the update launch's time goes, level by level of its dependent-load chain.

Runs the bench workload (FASTSLAM2, example_webmap, Philox, fast build) through the INSTRUMENTED library
(slam_amd/libslamgpu_stamps.so, `make -C slam_amd/csrc stamps`), in which thread 0 of every compute block drains its
wave's outstanding memory operations and records the 100 MHz wall clock at ten points of update_kernel.  After each of
`samples` steps (mid-run) the stamps of that launch are read back; the table gives, per level, the median / p90 / max
over blocks and launches of the time since the EARLIEST block entered the kernel, split by whether the launch had to
perform the previous step's resampling (inline plan fired) or not.

usage: python tools/stamps.py [N] [samples] [host|device] [FASTSLAM2|FASTSLAM1] [fast|strict|flow] [map]  (writes a table to stdout; copy into profiles/)
       map: a bundled map's name (default example_webmap), e.g. example_loop902
       strict: the strict build's kernels (`make -C slam_amd/csrc stamps_strict`)
       device: the steps are slamgpu_step_observe calls (the observation front end inside the update launch)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
STRICT = len(sys.argv) > 5 and sys.argv[5] == "strict"
FLOW = len(sys.argv) > 5 and sys.argv[5] == "flow"   # stamps without drains (make stamps_flow): the launch's own schedule
MAP = sys.argv[6] if len(sys.argv) > 6 else "example_webmap"
os.environ["SLAMGPU_LIB"] = os.path.join(ROOT, "slam_amd", "libslamgpu_stamps_strict.so" if STRICT else ("libslamgpu_stamps_flow.so" if FLOW else "libslamgpu_stamps.so"))
os.environ["SLAMGPU_STAMPS"] = "1"
import numpy as np  # noqa: E402
import slam_amd  # noqa: E402
from slam_amd import host  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
SAMPLES = int(sys.argv[2]) if len(sys.argv) > 2 else 200
DEVICE = len(sys.argv) > 3 and sys.argv[3] == "device"
METHOD = sys.argv[4] if len(sys.argv) > 4 else "FASTSLAM2"
MID = 2 if METHOD == "FASTSLAM2" else 1
START = 1000
LEVELS = ["0 kernel entry (first block = 0)", "1 Ctrl words arrived", "2 block totals scanned (W, Neff, decision)",
          "3 ancestor found", "4 pose + genealogy arrived", "5 records staged in LDS", "6 proposal pass done",
          "7 second pass done, record stores landed", "8 pose / genealogy stores landed", "9 weight prefix + totals written",
          "10 queued predicts applied", "11 -", "12 -", "13 -", "14 (free slot: wherever a diagnostic build puts SLAM_STAMP(14))",
          "15 (free slot: SLAM_STAMP(15))"]
ORDER = [0, 1, 2, 3, 4, 10, 5, 6, 7, 8, 14, 15, 9]   # the order the kernel passes them in (14, 15: only if stamped)

tape = host.make_tape(["-m", os.path.join(ROOT, "data", MAP + ".mat"), "-method", METHOD, "-NPARTICLES", N,
                       "-NEFFECTIVE", int(0.75 * N), "-SWITCH_SEED_RANDOM", 7], max_obs=START + SAMPLES + 2)
Q, R, dt = tape["Q"], tape["R"], float(tape["dt"])
_sim = host.HostSim(["-m", os.path.join(ROOT, "data", MAP + ".mat"), "-method", "FASTSLAM2"])
LM, _ = _sim.map()
MAX_RANGE = float(_sim.conf.MAX_RANGE)
_sim.close()


def make():
    s = slam_amd.SlamGpu(N, tape["nlm"], method=MID, n_effective=int(0.75 * N), rng_mode=slam_amd.RNG_PHILOX, seed=7, math_mode=0 if STRICT else 1, device_observe=DEVICE,
                        use_heading=bool(tape["conf"].SWITCH_HEADING_KNOWN), wheel_base=float(tape["conf"].WHEELBASE), sigma_phi=float(tape["conf"].sigmaT))
    if DEVICE:
        s.set_map(LM)
        calls = [s.prepare_step_observe(np.array(st["controls"], np.float32).reshape(-1, 3), Q, dt, st["true"], MAX_RANGE, R, noise=2) for st in tape["steps"]]
    else:
        calls = [s.prepare_step(np.array(st["controls"], np.float32).reshape(-1, 3), Q, dt, st["zf"], st["idf"], st["zn"], R) for st in tape["steps"]]
    return s, calls


s, calls = make()
for c in calls[:START]:
    c()
s.sync()
s.estimate_fetch()
rows = {True: [], False: []}
last = []
for k in range(START, START + SAMPLES):
    calls[k]()
    st = s.debug_stamps().astype(np.int64)   # synchronises: the launch of step k has finished
    last.append(st)
_, _, res = s.history_fetch()
# NB reading the stamps forces the resampling stage of step k to run as its own launch (history / sync), so the NEXT
# update launch does not plan inline.  To see the inline-plan path the stamps are also taken without synchronising in
# between: run two steps back to back and read the second launch's stamps.
s.close()

s, calls = make()
for c in calls[:START]:
    c()
s.sync()
s.estimate_fetch()
pairs = []
k = START
while k + 1 < START + SAMPLES:
    calls[k]()
    calls[k + 1]()                       # this launch plans step k's resampling inline
    st = s.debug_stamps().astype(np.int64)
    pairs.append((k + 1, st))
    k += 2
_, _, res2 = s.history_fetch()
s.close()
res2 = list(res2)


def table(title, stamp_sets):
    print("\n== %s (%d launches, %d blocks each); microseconds since the first block entered the kernel" % (title, len(stamp_sets), stamp_sets[0].shape[0] if stamp_sets else 0))
    if not stamp_sets:
        return
    rel = []
    for st in stamp_sets:
        t0 = st[:, 0].min()
        rel.append((st[:, :16] - t0) / 100.0)   # 100 MHz -> us
    rel = np.concatenate(rel)
    print("%-48s %8s %8s %8s   %s" % ("level", "median", "p90", "max", "median step from previous level"))
    prev = None
    for j in ORDER:
        name = LEVELS[j]
        col = rel[:, j]
        med = np.median(col)
        if med < 0 or med > 1e7:                # a level this kernel variant does not stamp (FASTSLAM1 has one pass)
            continue
        print("%-48s %8.2f %8.2f %8.2f   %s" % (name, med, np.quantile(col, 0.9), col.max(), "" if prev is None else "%+.2f" % (med - prev)))
        prev = med
    ends = np.array([(st[:, 9].max() - st[:, 0].min()) / 100.0 for st in stamp_sets])
    print("last block's end - first block's entry: median %.2f us, max %.2f us" % (np.median(ends), ends.max()))


print("observation front end: %s" % ("inside the update launch (slamgpu_step_observe)" if DEVICE else "host (slamgpu_step)"))
print("method %s" % METHOD)
print("N = %d particles, steps %d..%d of the %s run, %s build, instrumented (thread 0 of each block drains vmcnt/lgkmcnt at each stamp)" % (N, START, START + SAMPLES, MAP, "strict" if STRICT else ("fast, stamps WITHOUT drains (the time a point was reached, not the time its data had arrived)" if FLOW else "fast")))
table("update launches that do NOT plan inline (previous stage already ran: pose read at slot i or through keep[])", last)
fired = [st for (kk, st) in pairs if res2[kk - START - 1]]
quiet = [st for (kk, st) in pairs if not res2[kk - START - 1]]
table("inline plan, previous step did NOT resample (normalise only)", quiet)
table("inline plan, previous step RESAMPLED (scan + ancestor search + gather through the ancestor)", fired)
