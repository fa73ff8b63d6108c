#!/bin/bash
set -o pipefail
D=gpurun_out/ka; mkdir -p $D
python -c "import __graft_entry__ as g; g.build()" > $D/build.log 2>&1 || { echo BUILD FAILED; exit 1; }
for v in 0 1; do
  export HIP_FORCE_DEV_KERNARG=$v
  for cfg in 3 5 2; do
  timeout -k 10 600 python bench.py --config $cfg --no-strict --no-cpu-baseline --single-pass > $D/b_$v_$cfg.json 2> $D/b_$v_$cfg.err || { echo "$v rc=$?"; tail -3 $D/b_$v_$cfg.err; }
  python -c "
import json; j=json.loads(open('$D/b_$v_$cfg.json').read().strip().splitlines()[-1]); print('DEV_KERNARG=$v config $cfg value %.4g ms/step %.5f' % (j['value'], j['ms_per_step']))"
  done
done
