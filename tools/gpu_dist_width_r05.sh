#!/bin/bash
# GPU box, round 5: where the wall time of the logical-shard harness (tools/dist_width.py) goes -- the G = 2 / G = 8 outliers of
# profiles/dist_width_r04.txt.  For each G: the harness alone (wall time per step), then the same under rocprofv3 --kernel-trace:
# kernel durations AND the gaps between consecutive kernels on the stream (idle GPU = the host was the limit), and the clocks.
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for G in 1 2 4 8; do
  echo "== G=$G, no profiler"; python3 tools/dist_width.py 100096 $G
  (rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk" | head -4) || true
  D=gpurun_out/dw5_$G; rm -rf $D
  rocprofv3 --kernel-trace --output-format csv -d $D -- python3 tools/dist_width.py 100096 $G > $D.log 2>&1
  grep "us per step" $D.log | sed 's/^/   under the profiler: /'
  python3 - <<PY
import csv, glob, collections
rows = []
for f in glob.glob("$D/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-(400 * ($G + 1)):]   # the timed window: 400 steps of G update launches + the gather kernel
dur = collections.defaultdict(list); gap = []
for a, b in zip(rows, rows[1:]):
    gap.append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
for r in rows:
    dur[r["Kernel_Name"].split("(")[0][:60]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
import statistics as st
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
busy = sum(sum(v) for v in dur.values())
print("   timed window on the GPU: %.2f us per step span, %.2f us per step busy (%.0f %%); gaps between kernels: median %.2f us, p90 %.2f, max %.1f" % (
      span / 400e3, busy / 400e3, 100.0 * busy / span, st.median(gap) / 1e3, sorted(gap)[int(0.9 * len(gap))] / 1e3, max(gap) / 1e3))
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print("   %-62s %6d launches, average %.2f us (min %.2f, max %.2f)" % (k, len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3))
PY
done
