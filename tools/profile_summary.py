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
path = os.path.join(out, "summary_%s.txt" % tag)
open(path, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:60]))
