#!/bin/bash
# GPU box: BASELINE config 5 with two builds of the library on ONE box, alternating (usage: tools/gpu_ab_c5.sh <name of the
# other build: slam_amd/libslamgpu_<name>.so> [rounds]); prints ms per step of every repeat
OTHER=${1:-prev}; ROUNDS=${2:-2}
for r in $(seq $ROUNDS); do
  for v in this $OTHER; do
    if [ $v = this ]; then unset SLAMGPU_LIB; else export SLAMGPU_LIB=$PWD/slam_amd/libslamgpu_$v.so; fi
    python bench.py --config 5 --steps 20 --warmup 3 --single-pass --repeats 3 --no-cpu-baseline > gpurun_out/ab_c5_$v.json 2>> gpurun_out/ab_c5.err || echo "bench failed ($v)"
    python - <<PY
import json
d = json.loads(open("gpurun_out/ab_c5_$v.json").read().strip().splitlines()[-1])
print("config 5 [%-5s]: %.4f ms per step, repeats %s" % ("$v", d["ms_per_step"], [round(x, 4) for x in d["window_repeats"]["ms_per_step_all"]]))
PY
  done
done
