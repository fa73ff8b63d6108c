#!/bin/bash
# usage: tools/gpu_variants.sh "<bench args>" name1 name2 ...   (variant libraries built with `make variant NAME=...`)
set -o pipefail
D=gpurun_out/var; mkdir -p $D
ARGS=$1; shift
python -c "import __graft_entry__ as g; g.build()" > $D/build.log 2>&1 || { echo BUILD FAILED; exit 1; }
for v in base "$@"; do
  if [ $v = base ]; then unset SLAMGPU_LIB; else export SLAMGPU_LIB=$PWD/slam_amd/libslamgpu_$v.so; fi
  for rep in 1 2; do
  timeout -k 10 600 python bench.py $ARGS --no-strict --no-cpu-baseline --single-pass > $D/b_$v.json 2> $D/b_$v.err || { echo "$v rc=$?"; tail -3 $D/b_$v.err; }
  python -c "
import json; j=json.loads(open('$D/b_$v.json').read().strip().splitlines()[-1]); print('$v value %.4g ms/step %.5f' % (j['value'], j['ms_per_step']))"
  done
done
