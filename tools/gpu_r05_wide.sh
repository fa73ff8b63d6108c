#!/bin/bash
# GPU box: large single contexts (BASELINE config 4's workload on ONE GPU and smaller sets): update_kernel_wide (three waves per
# SIMD) against update_kernel (SLAMGPU_NO_WIDE=1).
# usage: bash tools/gpu_r05_wide.sh   (table on stdout)
run() {  # label, env..., -- bench args
    local label=$1; shift
    local envs=()
    while [ "$1" != "--" ]; do envs+=("$1"); shift; done
    shift
    env "${envs[@]}" python bench.py --single-pass --repeats 5 --no-cpu-baseline --no-also --no-strict "$@" > gpurun_out/wide_tmp.json 2>> gpurun_out/wide.err || { echo "$label: bench failed"; return; }
    python - "$label" "$*" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/wide_tmp.json").read().strip().splitlines()[-1])
wr = d.get("whole_run") or {}
print("%-34s %-52s %8.3f us per step   whole run %8.3f us" % (sys.argv[1], sys.argv[2], 1e3 * d["ms_per_step"], 1e3 * wr.get("ms_per_step", float("nan"))))
PY
}
for n in 250000 500000 1000000; do
  run "update_kernel" SLAMGPU_NO_WIDE=1 -- --config 4 --particles $n --steps 20 --warmup 5
  run "update_kernel_wide" X=1 -- --config 4 --particles $n --steps 20 --warmup 5
done
# FastSLAM 1 at 10^6 particles (30 spilled registers in the wide form)
run "update_kernel (FASTSLAM1)" SLAMGPU_NO_WIDE=1 -- --config 2 --particles 1000000 --steps 20 --warmup 5 --observe host
run "update_kernel_wide (FASTSLAM1)" X=1 -- --config 2 --particles 1000000 --steps 20 --warmup 5 --observe host
