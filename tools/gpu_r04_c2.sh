#!/bin/bash
# GPU box: config 2 on ONE box: this tree's library against round 3's (slam_amd/libslamgpu_r3.so, built from commit b4af56d),
# twice each, interleaved; then the level stamps of this tree's kernel
for r in 1 2; do
  for v in r4 r3; do
    if [ $v = r3 ]; then export SLAMGPU_LIB=$PWD/slam_amd/libslamgpu_r3.so; else unset SLAMGPU_LIB; fi
    python bench.py --config 2 --steps 200 --warmup 20 --single-pass --no-cpu-baseline --repeats 3 > gpurun_out/c2_$v.json 2>> gpurun_out/c2.err
    python -c "
import json; d=json.loads(open('gpurun_out/c2_$v.json').read().strip().splitlines()[-1]); print('config 2', '$v', d['ms_per_step']*1e3, 'us', d['window_repeats']['ms_per_step_all'])"
    python bench.py --config 2 --observe device --steps 200 --warmup 20 --single-pass --no-cpu-baseline --repeats 3 > gpurun_out/c2_dev_$v.json 2>> gpurun_out/c2.err
    python -c "
import json; d=json.loads(open('gpurun_out/c2_dev_$v.json').read().strip().splitlines()[-1]); print('config 2 --observe device', '$v', d['ms_per_step']*1e3, 'us', d['window_repeats']['ms_per_step_all'])"
  done
done
unset SLAMGPU_LIB
timeout -k 10 300 python tools/stamps.py 1000 200 host FASTSLAM1 > gpurun_out/levels_c2_host_r04.txt 2>> gpurun_out/levels.err; echo "stamps rc=$?"
