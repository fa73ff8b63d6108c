#!/bin/bash
set -o pipefail
D=gpurun_out/c24; mkdir -p $D
python -c "import __graft_entry__ as g; g.build()" > $D/build.log 2>&1 || { echo BUILD FAILED; exit 1; }
for cfg in 4 2; do
timeout -k 10 900 python bench.py --config $cfg > $D/b_$cfg.json 2> $D/b_$cfg.err || { echo "cfg $cfg rc=$?"; tail -3 $D/b_$cfg.err; }
python -c "
import json; j=json.loads(open('$D/b_$cfg.json').read().strip().splitlines()[-1]); print('config $cfg value %.4g ms/step %.5f frac %.3f' % (j['value'], j['ms_per_step'], j['roofline']['frac']), j['config'].get('strict_build'), (j.get('cpu_baseline') or {}).get('value'))"
done
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $D/b_driver.json 2> $D/b_driver.err; python -c "
import json; j=json.loads(open('$D/b_driver.json').read().strip().splitlines()[-1]); print('driver-style --steps 20 --warmup 5: value %.4g ms/step %.5f' % (j['value'], j['ms_per_step']), j['roofline']['traffic'], j['roofline'].get('traffic_nearest'))"
