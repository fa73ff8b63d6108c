#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel trace stats + FETCH_SIZE / WRITE_SIZE PMC passes of tools/profile.sh) into
gpurun_out/<dir>/summary_<tag>.txt and traffic_<tag>.json (both copied into profiles/ by hand).  Per-dispatch figures of
the update kernel are taken over the LAST `steps` dispatches (= bench.py's timed window; the dispatches before it
advance the filter and warm up)."""
import collections
import csv
import glob
import json
import os
import re
import sys

out, tag = sys.argv[1], sys.argv[2]
lines = []


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


bench = None
try:
    bench = json.loads(open(os.path.join(out, "bench_trace.json")).read().strip().splitlines()[-1])
except Exception as e:  # noqa: BLE001
    lines.append("!! no bench JSON line in bench_trace.json: %s" % e)
steps = bench["steps"] if bench else None

for f in find("trace/**/*kernel_stats.csv"):
    lines.append("== kernel stats, whole process (%s)" % os.path.relpath(f, out))
    for r in csv.DictReader(open(f)):
        lines.append("%-110s calls %7s total_ns %14s avg_ns %12s pct %6s min %10s max %10s" % (
            r.get("Name", "")[:110], r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage"), r.get("MinNs"), r.get("MaxNs")))

# per-dispatch durations from the kernel trace, in dispatch order
dur = collections.defaultdict(list)
for f in find("trace/**/*kernel_trace.csv"):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r.get("Start_Timestamp", 0)))
    for r in rows:
        dur[r.get("Kernel_Name", "")].append(int(r.get("End_Timestamp", 0)) - int(r.get("Start_Timestamp", 0)))


def dominant(names):
    cands = [k for k in names if re.search(r"update_kernel<|update_kernel_wide<|update_persist_kernel<", k)]
    return max(cands, key=lambda k: sum(dur.get(k, [0])) if dur else 0) if cands else None


dom = dominant(dur.keys())
# the persistent step loop (round 5): the timed window is ONE dispatch of `steps` iterations (the last one of the process): figures
# per ITERATION = that dispatch's / steps
persist = bool(dom) and "update_persist_kernel<" in dom
per_disp = steps if persist else 1
window_avg_ns = None
if dom and steps:
    w = dur[dom][-1:] if persist else dur[dom][-steps:]
    window_avg_ns = sum(w) / max(len(w), 1) / per_disp
    if persist:
        lines.append("== timed window: the LAST dispatch of %s = %d iterations: %.1f ns per iteration (%d ns the dispatch)" % (dom.split("(")[0], steps, window_avg_ns, w[0]))
    else:
        lines.append("== timed window: last %d dispatches of %s: avg %.1f ns (min %d, max %d)" % (len(w), dom.split("(")[0], window_avg_ns, min(w), max(w)))

pmc = {}
for name, pm in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    per = collections.defaultdict(list)
    for f in find(pm + "/**/*counter_collection.csv"):
        rows = [r for r in csv.DictReader(open(f)) if r.get("Counter_Name") == name]
        rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
        # one row per (dispatch, XCD/instance) in some rocprofv3 versions: sum per dispatch
        agg = collections.OrderedDict()
        for r in rows:
            key = (r.get("Kernel_Name", ""), int(r.get("Dispatch_Id", 0)))
            agg[key] = agg.get(key, 0.0) + float(r.get("Counter_Value", 0))
        for (k, _), v in agg.items():
            per[k].append(v)
    pmc[name] = per
    lines.append("== %s per dispatch [counter unit: KiB as reported by rocprofv3; gfx950: FETCH_SIZE counts 64 B per 128-B request => x2]" % name)
    for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        tail = (v[-1:] if persist else v[-steps:]) if (steps and k == dom) else v
        lines.append("%-110s dispatches %7d mean(all) %14.2f mean(window) %14.2f" % (k[:110], len(v), sum(v) / max(len(v), 1), sum(tail) / max(len(tail), 1)))

kern = {}
if dom:
    f_ = pmc["FETCH_SIZE"].get(dom, [])
    w_ = pmc["WRITE_SIZE"].get(dom, [])
    if persist:
        fk = f_[-1] / per_disp if f_ else 0.0
        wk = w_[-1] / per_disp if w_ else 0.0
    else:
        fk = sum(f_[-steps:]) / max(len(f_[-steps:]), 1) if f_ else 0.0
        wk = sum(w_[-steps:]) / max(len(w_[-steps:]), 1) if w_ else 0.0
    short = "fs2_update" if ("update_kernel<2" in dom or "update_kernel_wide<2" in dom or "update_persist_kernel<2" in dom) else "fs1_update"
    kern[short] = {"rocprof_name": dom.split("(")[0], "fetch_kib_mean": round(fk, 2), "write_kib_mean": round(wk, 2),
                   "hbm_bytes_per_launch": int((2 * fk + wk) * 1024), "avg_ns_rocprof": int(window_avg_ns) if window_avg_ns else None,
                   "dispatches_in_window": 1 if persist else steps, "iterations_per_dispatch": per_disp}
cfg = bench["config"] if bench else {}
json.dump({"_comment": "HBM-side traffic per dispatch of the dominant kernel over bench.py's timed window, from rocprofv3 PMC passes "
                       "(tools/profile.sh: separate passes for FETCH_SIZE and WRITE_SIZE, counter unit KiB).  gfx950 correction per "
                       "MI355X_MICROARCH.md: FETCH_SIZE counts 64 B per 128-B request => doubled.  Infinity-Cache hits are not excluded "
                       "by the fabric counters.",
           "source": "profiles/rocprof_summary_%s.txt" % tag, "config": cfg.get("baseline_config"), "math": cfg.get("math"),
           "particles_per_gpu": cfg.get("particles_per_gpu"), "workload": cfg.get("workload"), "steps": steps,
           "bench_avg_launch_us": (bench or {}).get("roofline", {}).get("avg_launch_us"), "kernels": kern},
          open(os.path.join(out, "traffic_%s.json" % tag), "w"), indent=1)
path = os.path.join(out, "summary_%s.txt" % tag)
open(path, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:80]))
