#!/bin/bash
# GPU box: config 2 A/B (hoisted predict noise on / off), level stamps of the new kernel
for v in hoist nohoist; do
  if [ $v = nohoist ]; then export SLAMGPU_NO_HOIST=1; else unset SLAMGPU_NO_HOIST; fi
  for r in 1 2; do
    python bench.py --config 2 --steps 200 --warmup 20 --single-pass --no-cpu-baseline --repeats 3 > gpurun_out/c2_$v.json 2>> gpurun_out/c2.err
    python -c "
import json; d=json.loads(open('gpurun_out/c2_$v.json').read().strip().splitlines()[-1]); print('$v', d['ms_per_step']*1e3, 'us', d['window_repeats']['ms_per_step_all'])"
  done
  timeout -k 10 300 python tools/stamps.py 1000 200 host FASTSLAM1 > gpurun_out/levels_c2_host_$v.txt 2>> gpurun_out/levels.err; echo "stamps rc=$?"
done
