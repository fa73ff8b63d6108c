#!/bin/bash
# round 5: example_loop902 (117 landmarks) at 10^5 particles: the compact layout against the plain rows of rounds 1-4, one box
set -e
out=gpurun_out/loop902_${1:-r05}
mkdir -p $out
common="--config 6 --steps 200 --warmup 20 --no-also --no-cpu-baseline --no-strict --single-pass"
python bench.py $common > $out/compact.json
SLAMGPU_NO_MID_COMPACT=1 python bench.py $common > $out/plain_rows.json
python bench.py $common --math strict > $out/compact_strict.json
SLAMGPU_NO_MID_COMPACT=1 python bench.py $common --math strict > $out/plain_rows_strict.json
python - <<PY
import json
for n in ("compact", "plain_rows", "compact_strict", "plain_rows_strict"):
    j = json.load(open("$out/%s.json" % n))
    w = j.get("whole_run", {})
    print("%-20s %8.3f us per step (window %s, m %.2f, rows in use %s)   whole run %8.3f us" % (n, 1e3 * j["ms_per_step"], j["config"]["window_start"],
          j["config"]["mean_m"], j["config"]["genealogy_rows_in_use"], 1e3 * w.get("ms_per_step", float("nan"))))
PY
