#!/usr/bin/env python3
"""Diagnostic (GPU box): distribution of GPU-vs-oracle deviations per quantity, teacher-forced per step."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import slam_amd
from oracle import orc
from test_gpu_parity import drive_pair
import fs2_float64

def main():
    O = orc.Oracle()
    for method, N, seed, nobs in [("FASTSLAM2", 100, 7, 400), ("FASTSLAM2", 1000, 1, 100), ("FASTSLAM1", 100, 7, 200)]:
        for mm in (0, 1):
            rw, rx, rxf, rn, anc_bad, anc_tot, nres = [], [], [], [], 0, 0, 0
            t_gpu, t_ref = [], []   # normalised weights against the float64 evaluation of the same update
            def per(r):
                nonlocal anc_bad, anc_tot, nres
                if r["did"][0] != r["did"][1]:
                    print("  DECISION MISMATCH at", r["k"], r["neff"]); return
                rn.append(abs(r["neff"][0]-r["neff"][1])/r["neff"][1])
                if r["did"][0]:
                    nres += 1
                    bad = np.abs(r["got"]["xv"] - r["exp"]["xv"]).max(axis=1) > 2e-4
                    anc_bad += bad.sum(); anc_tot += bad.size
                else:
                    rw.append(np.abs(r["got"]["w"]/r["exp"]["w"] - 1))
                    if method == "FASTSLAM2" and r["m"] > 0:
                        _, wt = fs2_float64.update_weights(r["pre"], r["obs"]["zf"], r["obs"]["idf"], r["R"], r["normals"])
                        wt = wt / wt.sum()
                        t_gpu.append(np.abs(r["got"]["w"].astype(np.float64) / r["got"]["w"].sum(dtype=np.float64) / wt - 1))
                        t_ref.append(np.abs(r["exp"]["w"].astype(np.float64) / r["exp"]["w"].sum(dtype=np.float64) / wt - 1))
                    rx.append(np.abs(r["got"]["xv"] - r["exp"]["xv"]).max(axis=1))
                    if r["got"]["nf"]:
                        rxf.append(np.abs(r["got"]["xf"] - r["exp"]["xf"]).max())
            drive_pair(slam_amd, O, "example_webmap", method, N, seed, nobs, math_mode=mm, per_step=per, want_pre=True)
            rw = np.concatenate(rw) if rw else np.zeros(1); rx = np.concatenate(rx) if rx else np.zeros(1)
            q = lambda a: "med %.2e p99 %.2e max %.2e" % (np.median(a), np.quantile(a, 0.99), np.max(a))
            print(method, N, "math", ["strict","fast"][mm], "| w rel:", q(rw), "| xv abs:", q(rx), "| xf max %.2e" % (max(rxf) if rxf else 0),
                  "| neff rel max %.2e" % max(rn), "| resample steps", nres, "ancestor-mismatch frac %.4f" % (anc_bad/max(anc_tot,1)))
            if t_gpu:
                print("    vs float64 evaluation, normalised weights:  GPU", q(np.concatenate(t_gpu)), " | float32 reference", q(np.concatenate(t_ref)))
main()
