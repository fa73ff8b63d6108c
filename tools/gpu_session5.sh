#!/bin/bash
set -o pipefail
D=gpurun_out/s5
mkdir -p $D
python -c "import __graft_entry__ as g; g.build()" > $D/build.log 2>&1 || { echo BUILD FAILED; tail -20 $D/build.log; exit 1; }
make -s -C slam_amd/csrc stamps >> $D/build.log 2>&1
timeout -k 10 1100 python -m pytest tests -m gpu -q --timeout 900 > $D/gputests.log 2>&1; echo "pytest rc=$?"
tail -12 $D/gputests.log
timeout -k 10 300 python bench.py --no-cpu-baseline > $D/bench_c3.json 2> $D/bench_c3.err; echo "c3 rc=$?"; cat $D/bench_c3.json
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strict > $D/bench_c3_short.json 2> $D/bench_c3_short.err; echo "c3 short rc=$?"; cat $D/bench_c3_short.json
timeout -k 10 600 python bench.py --config 5 --no-strict --no-cpu-baseline > $D/bench_c5.json 2> $D/bench_c5.err; echo "c5 rc=$?"; cat $D/bench_c5.json; tail -3 $D/bench_c5.err
timeout -k 10 300 python tools/stamps.py 100000 200 > $D/stamps_N100000.txt 2>&1; echo "stamps rc=$?"; tail -28 $D/stamps_N100000.txt
