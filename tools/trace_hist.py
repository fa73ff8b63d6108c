#!/usr/bin/env python3
"""Per-kernel duration quantiles from a rocprofv3 --kernel-trace CSV (the update kernel is bimodal: launches that
carry a lazy gather and launches that do not).  usage: tools/trace_hist.py <dir>"""
import csv, glob, os, sys, collections
import numpy as np
d = sys.argv[1]
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    v = np.array(v)
    if len(v) < 20:
        continue
    q = np.quantile(v, [0.05, 0.25, 0.5, 0.75, 0.95])
    print("%-70s n %6d mean %7.2f us | p5 %6.2f p25 %6.2f p50 %6.2f p75 %6.2f p95 %6.2f" % (k[:70], len(v), v.mean(), *q))
