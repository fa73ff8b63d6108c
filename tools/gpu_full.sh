#!/bin/bash
set -o pipefail
D=gpurun_out/full; mkdir -p $D
python -c "import __graft_entry__ as g; g.build()" > $D/build.log 2>&1 || { echo BUILD FAILED; tail $D/build.log; exit 1; }
timeout -k 10 1100 python -m pytest tests -m gpu -q --timeout 900 > $D/gputests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $D/gputests.log
[ $rc -eq 0 ] || exit $rc
for cfg in 3 5 2; do
timeout -k 10 600 python bench.py --config $cfg --no-strict --no-cpu-baseline --single-pass > $D/b_$cfg.json 2> $D/b_$cfg.err || { echo "cfg $cfg rc=$?"; tail -3 $D/b_$cfg.err; }
python -c "
import json; j=json.loads(open('$D/b_$cfg.json').read().strip().splitlines()[-1]); print('config $cfg value %.4g ms/step %.5f' % (j['value'], j['ms_per_step']))"
done
