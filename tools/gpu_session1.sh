#!/bin/bash
# GPU box session: GPU test-suite, then bench lines (output under gpurun_out/)
set -o pipefail
mkdir -p gpurun_out/s1
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/s1/build.log 2>&1 || { echo BUILD FAILED; tail -20 gpurun_out/s1/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 > gpurun_out/s1/gputests.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/s1/gputests.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/s1/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/s1/smoke.log
timeout -k 10 300 python bench.py > gpurun_out/s1/bench_c3.json 2> gpurun_out/s1/bench_c3.err; echo "bench rc=$?"; cat gpurun_out/s1/bench_c3.json
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/s1/bench_c3_short.json 2> gpurun_out/s1/bench_c3_short.err; echo "short rc=$?"; cat gpurun_out/s1/bench_c3_short.json
timeout -k 10 200 python bench.py --config 2 --no-cpu-baseline > gpurun_out/s1/bench_c2.json 2> gpurun_out/s1/bench_c2.err; echo "c2 rc=$?"; cat gpurun_out/s1/bench_c2.json
timeout -k 10 300 python bench.py --config 4 --no-cpu-baseline > gpurun_out/s1/bench_c4.json 2> gpurun_out/s1/bench_c4.err; echo "c4 rc=$?"; cat gpurun_out/s1/bench_c4.json
