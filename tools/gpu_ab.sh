#!/bin/bash
# GPU box: one bench workload with two builds of the library on ONE box, alternating
# usage: tools/gpu_ab.sh <other build: slam_amd/libslamgpu_<name>.so> <rounds> <bench args...>
OTHER=${1:-prev}; ROUNDS=${2:-2}; shift 2
for r in $(seq $ROUNDS); do
  for v in this $OTHER; do
    if [ $v = this ]; then unset SLAMGPU_LIB; else export SLAMGPU_LIB=$PWD/slam_amd/libslamgpu_$v.so; fi
    python bench.py --single-pass --repeats 5 --no-cpu-baseline "$@" > gpurun_out/ab_$v.json 2>> gpurun_out/ab.err || echo "bench failed ($v)"
    python - <<PY
import json
d = json.loads(open("gpurun_out/ab_$v.json").read().strip().splitlines()[-1])
print("[%-5s] %s: %.3f us per step, repeats %s" % ("$v", "$*", 1e3 * d["ms_per_step"], [round(1e3 * x, 3) for x in d["window_repeats"]["ms_per_step_all"]]))
PY
  done
done
