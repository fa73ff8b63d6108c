#!/bin/bash
D=gpurun_out/s6; mkdir -p $D
for v in 0 1; do
  HIP_FORCE_DEV_KERNARG=$v timeout -k 10 200 python bench.py --no-cpu-baseline --no-strict --single-pass > $D/bench_kernarg$v.json 2> $D/bench_kernarg$v.err; echo "HIP_FORCE_DEV_KERNARG=$v rc=$?"
  python -c "
import json; j=json.loads(open('$D/bench_kernarg$v.json').read().strip().splitlines()[-1]); print('  value %.4g ms/step %.5f' % (j['value'], j['ms_per_step']))"
done
HIP_FORCE_DEV_KERNARG=1 timeout -k 10 300 python tools/stamps.py 100000 100 > $D/stamps_devkernarg.txt 2>&1; tail -14 $D/stamps_devkernarg.txt
