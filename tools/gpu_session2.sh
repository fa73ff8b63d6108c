#!/bin/bash
# GPU box session 2: new tests (log-weights, config 5, KATs, backend), config-5 bench
set -o pipefail
D=gpurun_out/s2
mkdir -p $D
python -c "import __graft_entry__ as g; g.build()" > $D/build.log 2>&1 || { echo BUILD FAILED; tail -20 $D/build.log; exit 1; }
timeout -k 10 300 python tools/debug_backend.py > $D/debug_backend.log 2>&1; echo "debug rc=$?"; tail -12 $D/debug_backend.log
timeout -k 10 1000 python -m pytest tests -m gpu -q --timeout 900 -x -k "kat or logweights or config5 or backend" > $D/gputests_new.log 2>&1; echo "pytest(new) rc=$?"
tail -25 $D/gputests_new.log
timeout -k 10 600 python bench.py --config 5 --no-strict --cpu-seconds 16 > $D/bench_c5.json 2> $D/bench_c5.err; echo "c5 rc=$?"; cat $D/bench_c5.json; tail -5 $D/bench_c5.err
