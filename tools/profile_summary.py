#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel trace stats + FETCH_SIZE / WRITE_SIZE PMC passes) into a small text summary
under gpurun_out/<dir>/summary_<tag>.txt (copied into profiles/ by hand)."""
import csv, glob, os, sys, collections
out, tag = sys.argv[1], sys.argv[2]
lines = []
def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))
# kernel stats
for f in find("trace/**/*kernel_stats.csv"):
    lines.append("== kernel stats (%s)" % os.path.relpath(f, out))
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        lines.append("%-110s calls %7s total_ns %12s avg_ns %10s pct %6s min %8s max %8s" % (
            r.get("Name", "")[:110], r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage"), r.get("MinNs"), r.get("MaxNs")))
# per-dispatch trace: avg duration per kernel over the LAST half of dispatches is not needed; keep totals only
for name, pm in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    for f in find(pm + "/**/*counter_collection.csv"):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != name:
                continue
            k = r.get("Kernel_Name", "")
            agg[k][0] += float(r.get("Counter_Value", 0))
            agg[k][1] += 1
        lines.append("== %s per dispatch (%s) [counter unit: KiB as reported by rocprofv3; gfx950: FETCH_SIZE counts 64 B per 128-B request => x2]" % (name, os.path.relpath(f, out)))
        for k, (v, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
            lines.append("%-110s dispatches %7d mean %14.2f" % (k[:110], n, v / max(n, 1)))
# machine-readable HBM traffic per launch for bench.py's roofline.traffic (gfx950: FETCH_SIZE x2, unit KiB)
import json, re
means = {"FETCH_SIZE": {}, "WRITE_SIZE": {}}
for name, pm in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    for f in find(pm + "/**/*counter_collection.csv"):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == name:
                agg[r.get("Kernel_Name", "")][0] += float(r.get("Counter_Value", 0)); agg[r.get("Kernel_Name", "")][1] += 1
        for k, (v, n) in agg.items():
            means[name][k] = v / max(n, 1)
avg_ns = {}
for f in find("trace/**/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        avg_ns[r.get("Name", "")] = float(r.get("AverageNs", 0))
math = None
try:
    math = json.loads(open(os.path.join(out, "bench_trace.json")).read().strip().splitlines()[-1])["config"]["math"]
except Exception:
    pass
kern = {}
for short, pat in (("fs2_update", r"update_kernel<2"), ("fs1_update", r"update_kernel<1"), ("resample", r"resample_kernel"), ("finish", r"finish_kernel"), ("gather", r"gather_kernel")):
    for k in means["FETCH_SIZE"]:
        if re.search(pat, k):
            fk, wk = means["FETCH_SIZE"].get(k, 0.0), means["WRITE_SIZE"].get(k, 0.0)
            an = [v for n, v in avg_ns.items() if re.search(pat, n)]
            kern[short] = {"rocprof_name": k.split("(")[0], "fetch_kib_mean": round(fk, 2), "write_kib_mean": round(wk, 2),
                           "hbm_bytes_per_launch": int((2 * fk + wk) * 1024), "avg_ns_rocprof": int(an[0]) if an else None}
json.dump({"_comment": "HBM-side traffic per dispatch from rocprofv3 PMC passes of the default bench.py run (tools/profile.sh; separate passes "
                       "for FETCH_SIZE and WRITE_SIZE, counter unit KiB). gfx950 correction per MI355X_MICROARCH.md: FETCH_SIZE counts 64 B "
                       "per 128-B request => doubled. The particle state (2 x 88 MB) is Infinity-Cache resident at this size; the fabric "
                       "counters do not exclude such hits.",
           "source": "profiles/rocprof_summary_%s.txt" % tag, "math": math, "kernels": kern},
          open(os.path.join(out, "traffic_%s.json" % tag), "w"), indent=1)
path = os.path.join(out, "summary_%s.txt" % tag)
open(path, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:60]))
