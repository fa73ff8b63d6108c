#!/bin/bash
set -o pipefail
D=gpurun_out/c5b; mkdir -p $D
python -c "import __graft_entry__ as g; g.build()" > $D/build.log 2>&1 || { echo BUILD FAILED; exit 1; }
make -s -C slam_amd/csrc stamps > $D/stamps_build.log 2>&1 || { echo STAMPS BUILD FAILED; tail $D/stamps_build.log; exit 1; }
timeout -k 10 600 python -m pytest tests -m gpu -q --timeout 900 -x -k "config5 or logweights or many_landmarks or lazy_gather or dist" > $D/tests.log 2>&1; echo "tests rc=$?"; tail -3 $D/tests.log
timeout -k 10 600 python bench.py --config 5 --no-strict --no-cpu-baseline --single-pass > $D/bench_c5.json 2> $D/bench_c5.err; echo "c5 rc=$?"
python -c "
import json; j=json.loads(open('$D/bench_c5.json').read().strip().splitlines()[-1]); print('  c5 value %.4g ms/step %.4f' % (j['value'], j['ms_per_step']))"
timeout -k 10 600 python tools/stamps_c5.py 8 > $D/stamps_c5.txt 2> $D/stamps_c5.err; echo "rc=$?"; tail -4 $D/stamps_c5.txt
