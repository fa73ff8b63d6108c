#!/bin/bash
# GPU box: config 5 against the geometry of the landmark pipeline (chunk size x chunks in flight); variants built by
# `make -C slam_amd/csrc variant NAME=.. EXTRA=..`, loaded through SLAMGPU_LIB
for v in base c4d2 c4d3 c8d1 c8d2 c2d3 base; do
  if [ $v = base ]; then unset SLAMGPU_LIB; else export SLAMGPU_LIB=$PWD/slam_amd/libslamgpu_$v.so; fi
  python bench.py --config 5 --steps 20 --warmup 3 --single-pass --no-cpu-baseline --repeats 3 > gpurun_out/c5_$v.json 2>> gpurun_out/c5.err
  python -c "
import json; d=json.loads(open('gpurun_out/c5_$v.json').read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], 'ms', d['window_repeats']['ms_per_step_all'])"
done
