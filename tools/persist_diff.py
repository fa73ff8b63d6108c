"""diagnostic: per-step launches vs the persistent loop, first differing step and field (example_loop2, FS2, fast build)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import slam_amd as sg
from slam_amd import host
from conftest import sim_args
mapname, method, N, math = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
args = sim_args(mapname, method, 100, 7)
tape = host.make_tape(args, max_obs=12)
sim = host.HostSim(args); lm, _ = sim.map(); mr = float(sim.conf.MAX_RANGE); sim.close()
steps = tape["steps"]; conf = tape["conf"]
ctl = [np.array(st["controls"], np.float32).reshape(-1, 3) for st in steps]
xt = [np.asarray(st["true"], np.float32) for st in steps]
def make():
    s = sg.SlamGpu(N, tape["nlm"], method=2 if method == "FASTSLAM2" else 1, n_effective=int(0.75 * N), rng_mode=sg.RNG_PHILOX, seed=5,
                   math_mode=math, device_observe=True, use_heading=bool(conf.SWITCH_HEADING_KNOWN), wheel_base=float(conf.WHEELBASE), sigma_phi=float(conf.sigmaT))
    s.set_map(lm); return s
for K in range(2, 8):
    a = make(); b = make()
    a.run_observe(ctl[:K], tape["Q"], float(tape["dt"]), xt[:K], mr, tape["R"], noise=2)
    for c, x in zip(ctl[:K], xt[:K]): b.step_observe(c, tape["Q"], float(tape["dt"]), x, mr, tape["R"], noise=2)
    da, db = a.download(), b.download()
    bad = False
    for key in ("xv", "Pv", "w", "xf", "Pf"):
        ua, ub = da[key].view(np.uint32), db[key].view(np.uint32)
        if not np.array_equal(ua, ub):
            idx = np.argwhere(ua != ub)
            print("K=%d %s differs at %d entries; first %s: %r vs %r" % (K, key, len(idx), idx[:4].tolist(), da[key][tuple(idx[0])], db[key][tuple(idx[0])]))
            bad = True
    print("K=%d nf=%d m=%s persist=%s %s" % (K, da["nf"], [st["zf"].shape[0] for st in steps[:K]], a.persist_info(), "DIFF" if bad else "same"))
    a.close(); b.close()
    if bad: break
