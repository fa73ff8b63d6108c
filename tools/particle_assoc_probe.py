#!/usr/bin/env python3
"""Whole runs with the PER-PARTICLE association carried into the update (slamgpu_update_particle): seeds x builds x particle counts
on a bundled map, beside the same run with the reference's known association (dataAssociationKnown, core.cpp:91-120).  Reported per
run: landmarks in the map of the best (largest-weight) particle, how many true landmarks that map covers within 1 m and how many of
its entries lie further than 1 m from every true landmark, slots in use / dead at the end, the largest number of slots ever rewritten
in one step, mean position error of the estimate, mean error of the known-association twin, milliseconds per observation step.
A run is GOOD when the best particle's map has at most --max-landmarks entries and the mean position error is under 1 m.
usage: tools/particle_assoc_probe.py [--seeds 7-16] [--particles 512,2048] [--map example_webmap] [--new-share 0.02] [--p-new auto]
GPU box, one process."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data")


def sim_args(mapname, N, seed, method="FASTSLAM2"):
    return ["-m", os.path.join(DATA, mapname + ".mat"), "-method", method, "-NPARTICLES", N, "-NEFFECTIVE", int(0.75 * N), "-SWITCH_SEED_RANDOM", seed]


def run(sg, host, mapname, N, seed, math, a, known):
    f32 = np.float32
    tape = host.make_tape(sim_args(mapname, N, seed, a.method))
    sim = host.HostSim(sim_args(mapname, N, seed, a.method))
    lm, _ = sim.map()
    sim.close()
    R = tape["R"]
    p_new = a.p_new if a.p_new > 0 else float(np.exp(-0.5 * a.gate_reject) / (2 * np.pi * np.sqrt(np.linalg.det(np.asarray(R, np.float64)))))
    cap = tape["nlm"] if known else a.slots * tape["nlm"]
    s = sg.SlamGpu(N, cap, method=2 if a.method == "FASTSLAM2" else 1, n_effective=int(0.75 * N), rng_mode=sg.RNG_PHILOX, seed=seed, math_mode=math,
                   particle_maps=not known, use_heading=bool(tape["conf"].SWITCH_HEADING_KNOWN), wheel_base=float(tape["conf"].WHEELBASE),
                   sigma_phi=float(tape["conf"].sigmaT))
    errs, most, dropped = [], 0, 0
    t0 = time.perf_counter()
    for st in tape["steps"]:
        for V, G, phi in np.array(st["controls"], f32).reshape(-1, 3):
            s.predict(float(V), float(G), tape["Q"], float(tape["dt"]), float(phi))
        zf, zn = np.array(st["zf"], f32).reshape(-1, 2), np.array(st["zn"], f32).reshape(-1, 2)
        if len(zf) + len(zn):
            if known:
                s.update(zf, np.array(st["idf"], np.int32), zn, R)
            else:
                rep = s.update_particle(np.concatenate([zf, zn]), R, a.gate_reject, a.gate_augment, new_share=a.new_share, p_new=p_new, census_every=a.census,
                                        excl=(a.excl_base, a.excl_per_m, a.unique_ratio))
                most = max(most, rep["rewritten"])
                dropped += rep["dropped"]
        e = s.estimate()
        errs.append(float(np.hypot(e[0] - st["true"][0], e[1] - st["true"][1])))
    s.sync()
    ms = (time.perf_counter() - t0) * 1e3 / max(len(tape["steps"]), 1)
    d = s.download()
    rep_end = dict(slots=d["nf"])
    s.close()
    best = int(np.argmax(d["w"]))
    xf = d["xf"][best][: d["nf"]]
    held = xf[~np.isnan(xf[:, 0])]
    dist = np.hypot(held[:, None, 0] - lm[0][None, :], held[:, None, 1] - lm[1][None, :]) if len(held) else np.zeros((0, lm.shape[1]))
    covered = int((dist.min(axis=0) < 1.0).sum()) if len(held) else 0
    stray = int((dist.min(axis=1) >= 1.0).sum()) if len(held) else 0
    holders = (~np.isnan(d["xf"][:, : d["nf"], 0])).sum(axis=0) if d["nf"] else np.zeros(0, int)
    return dict(n_map=len(held), covered=covered, stray=stray, slots=rep_end["slots"], dead=int((holders == 0).sum()), most=most, dropped=dropped,
                err=float(np.mean(errs)), emax=float(np.max(errs)), ms=ms, nlm=lm.shape[1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="7-16")
    ap.add_argument("--particles", default="512,2048")
    ap.add_argument("--map", default="example_webmap")
    ap.add_argument("--method", default="FASTSLAM2")
    ap.add_argument("--builds", default="fast,strict")
    ap.add_argument("--gate-reject", type=float, default=4.0)
    ap.add_argument("--gate-augment", type=float, default=25.0)
    ap.add_argument("--new-share", type=float, default=0.02)
    ap.add_argument("--p-new", type=float, default=0.0, help="0: the Gaussian's value at the reject gate, exp(-gate_reject / 2) / (2 pi sqrt det R)")
    ap.add_argument("--census", type=int, default=1)
    ap.add_argument("--excl-base", type=float, default=2.0)
    ap.add_argument("--excl-per-m", type=float, default=0.05)
    ap.add_argument("--unique-ratio", type=float, default=2.0)
    ap.add_argument("--slots", type=int, default=4, help="slot capacity as a multiple of the map's landmarks")
    ap.add_argument("--max-landmarks", type=int, default=45)
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    import slam_amd as sg
    from slam_amd import host
    lo, hi = (int(x) for x in a.seeds.split("-"))
    good = total = like = 0
    em, ek = [], []
    for N in (int(x) for x in a.particles.split(",")):
        for seed in range(lo, hi + 1):
            for build in a.builds.split(","):
                math = 1 if build == "fast" else 0
                r = run(sg, host, a.map, N, seed, math, a, False)
                k = run(sg, host, a.map, N, seed, math, a, True)
                ok = r["n_map"] <= a.max_landmarks and r["err"] < 1.0
                ok_rel = r["n_map"] <= a.max_landmarks and r["err"] <= 1.2 * k["err"] + 0.05
                good += ok
                like += ok_rel
                total += 1
                em.append(r["err"])
                ek.append(k["err"])
                print("%-14s N=%5d seed %2d %-6s best particle's map %3d of %d (covers %2d, stray %2d)  slots %3d (dead %3d, most rewritten in a step %2d, dropped %d)  "
                      "mean err %.3f max %.3f  known-association %.3f  %.2f ms/step (known %.2f)  %s %s"
                      % (a.tag or a.map, N, seed, build, r["n_map"], r["nlm"], r["covered"], r["stray"], r["slots"], r["dead"], r["most"], r["dropped"], r["err"], r["emax"],
                         k["err"], r["ms"], k["ms"], "good" if ok else "BAD", "like-known" if ok_rel else "WORSE-THAN-KNOWN"), flush=True)
    print("%-14s GOOD %d of %d (best particle's map within --max-landmarks and mean position error < 1 m); %d of %d within 1.2 x + 0.05 m of the known-association twin; "
          "mean of the mean errors %.3f m (known association %.3f m); new_share %.3g gates %.3g / %.3g"
          % (a.tag or a.map, good, total, like, total, float(np.mean(em)), float(np.mean(ek)), a.new_share, a.gate_reject, a.gate_augment), flush=True)


if __name__ == "__main__":
    sys.exit(main())
