#!/bin/bash
# quick correctness subset + A/B of variant libraries.  usage: tools/gpu_ab.sh "<bench args>" variants...
set -o pipefail
D=gpurun_out/ab; mkdir -p $D
ARGS=$1; shift
python -c "import __graft_entry__ as g; g.build()" > $D/build.log 2>&1 || { echo BUILD FAILED; exit 1; }
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 900 -x -k "stepwise or full_size or lazy_gather or dist or sharded_steps or config5 or edges" > $D/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $D/tests.log
[ $rc -eq 0 ] || exit $rc
for v in base "$@"; do
  if [ $v = base ]; then unset SLAMGPU_LIB; else export SLAMGPU_LIB=$PWD/slam_amd/libslamgpu_$v.so; fi
  for rep in 1 2; do
  timeout -k 10 600 python bench.py $ARGS --no-strict --no-cpu-baseline --single-pass > $D/b_$v.json 2> $D/b_$v.err || { echo "$v rc=$?"; tail -3 $D/b_$v.err; }
  python -c "
import json; j=json.loads(open('$D/b_$v.json').read().strip().splitlines()[-1]); print('$v value %.4g ms/step %.5f' % (j['value'], j['ms_per_step']))"
  done
done
