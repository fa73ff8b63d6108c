#!/usr/bin/env python3
"""GPU box diagnostic: per-level wall clock of the config-5 update launch (BIG kernel: 10^5 particles, ~1.3 k re-observed
landmarks per step) through the instrumented library (make -C slam_amd/csrc stamps).  usage: python tools/stamps_c5.py [steps] [particles]
Per step: the levels (median over blocks) and, per block, how long the proposal pass (level 4 -> 6) and the second pass
(6 -> 7) took: median / p90 / max over the blocks, and separately over the blocks that entered late (the second block of a CU)."""
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SLAMGPU_LIB"] = os.path.join(ROOT, "slam_amd", "libslamgpu_stamps.so")
os.environ["SLAMGPU_STAMPS"] = "1"
import numpy as np  # noqa: E402
import bench  # noqa: E402
import slam_amd  # noqa: E402
from slam_amd import host  # noqa: E402

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 12
cfg = bench.CONFIGS[5]
N = int(sys.argv[2]) if len(sys.argv) > 2 else cfg["particles"]
tape = host.make_tape(bench.tape_args_for(cfg, N, tempfile.mkdtemp(prefix="slam_st5_")), max_obs=STEPS + 2)
Q, R, dt = tape["Q"], tape["R"], float(tape["dt"])
s = slam_amd.SlamGpu(N, tape["nlm"], method=2, n_effective=int(0.75 * N), rng_mode=slam_amd.RNG_PHILOX, seed=7, math_mode=1, log_weights=True)
LEVELS = ["0 entry", "1 ctrl", "2 scan", "3 ancestor", "4 pose+gen", "5 -", "6 proposal pass done", "7 second pass done", "8 pose/gen stores",
          "9 end"]
for k, st in enumerate(tape["steps"][:STEPS]):
    s.step(np.array(st["controls"], np.float32).reshape(-1, 3), Q, dt, st["zf"], st["idf"], st["zn"], R)
    x = s.debug_stamps().astype(np.int64)
    t0 = x[:, 0].min()
    rel = (x[:, :10] - t0) / 100.0
    med = np.median(rel, axis=0)
    print("step %2d m=%4d n=%4d | " % (k, st["zf"].shape[0], st["zn"].shape[0]) + "  ".join("%s %.0f" % (LEVELS[j].split()[0], med[j]) for j in (0, 2, 3, 4, 6, 7, 8, 9)) +
          " | last end %.0f us; entry spread p50 %.0f max %.0f" % ((x[:, 9].max() - t0) / 100.0, np.median(rel[:, 0]), rel[:, 0].max()))
    p1, p2 = rel[:, 6] - rel[:, 4], rel[:, 7] - rel[:, 6]
    q = lambda a: "%.0f / %.0f / %.0f" % (np.median(a), np.quantile(a, 0.9), a.max())
    print("        per block: proposal pass %s us, second pass %s us (median / p90 / max over %d blocks); end of pass 1 p10 %.0f p50 %.0f p90 %.0f; end of pass 2 p10 %.0f p50 %.0f p90 %.0f" % (
        q(p1), q(p2), rel.shape[0], np.quantile(rel[:, 6], 0.1), np.median(rel[:, 6]), np.quantile(rel[:, 6], 0.9), np.quantile(rel[:, 7], 0.1), np.median(rel[:, 7]), np.quantile(rel[:, 7], 0.9)))
s.close()
