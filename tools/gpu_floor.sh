#!/bin/bash
set -o pipefail
D=gpurun_out/floor; mkdir -p $D
python -c "import __graft_entry__ as g; g.build()" > $D/build.log 2>&1 || { echo BUILD FAILED; exit 1; }
make -s -C slam_amd/csrc stamps > $D/stamps_build.log 2>&1 || { echo STAMPS BUILD FAILED; exit 1; }
timeout -k 10 300 python tools/stamps.py 1024 200 > $D/levels_N1024.txt 2> $D/levels_N1024.err; echo "rc=$?"
tail -14 $D/levels_N1024.txt
for n in 1024 25600 100000; do
timeout -k 10 300 python bench.py --particles $n --no-strict --no-cpu-baseline --single-pass > $D/b_$n.json 2> $D/b_$n.err; python -c "
import json; j=json.loads(open('$D/b_$n.json').read().strip().splitlines()[-1]); print('N=$n ms/step %.5f' % j['ms_per_step'])"
done
