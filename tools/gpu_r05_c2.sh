#!/bin/bash
# round 5: BASELINE config 2 (FASTSLAM1, 1 000 particles) on one box: the persistent step loop (slamgpu_run_observe, one launch per
# batch) against the loops of launches.  Usage (GPU box): bash tools/gpu_r05_c2.sh [tag]
set -e
tag=${1:-r05}
out=gpurun_out/c2_$tag
mkdir -p $out
common="--config 2 --steps 200 --warmup 20 --no-also --no-cpu-baseline --no-strict --single-pass"
python bench.py $common --observe batched > $out/batched.json
SLAMGPU_NO_PERSIST=1 python bench.py $common --observe batched > $out/batched_no_persist.json
python bench.py $common --observe device > $out/device.json
python bench.py $common --observe host > $out/host.json
python - <<PY
import json
for n in ("batched", "batched_no_persist", "device", "host"):
    j = json.load(open("$out/%s.json" % n))
    w = j.get("whole_run", {})
    print("%-20s %8.3f us per step (window %s, repeats %s)   whole run %8.3f us" % (n, 1e3 * j["ms_per_step"], j["config"]["window_start"],
          ["%.2f" % (1e3 * x) for x in j["window_repeats"]["ms_per_step_all"]], 1e3 * w.get("ms_per_step", float("nan"))))
PY
