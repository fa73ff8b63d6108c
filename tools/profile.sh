#!/bin/bash
# GPU box: rocprofv3 kernel-trace stats + two separate PMC passes (FETCH_SIZE / WRITE_SIZE cannot share a pass on
# gfx950: TCC has 4 slots, FETCH_SIZE costs 3, WRITE_SIZE 2 -- MI355X_MICROARCH.md section rocprofv3 PMC slots) over
# `python3 bench.py --single-pass <bench args>`; tools/profile_summary.py condenses them over the LAST `steps` dispatches
# of the dominant kernel (= the timed window).
# usage: tools/profile.sh <tag> [bench args...]      e.g.  tools/profile.sh r02_c5 --config 5 --no-cpu-baseline
set -o pipefail
TAG=${1:-r02}; shift
ARGS=${@:---no-cpu-baseline}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --single-pass --repeats 1 $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err || echo "trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --single-pass --repeats 1 $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err || echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --single-pass --repeats 1 $ARGS > $OUT/bench_write.json 2> $OUT/write.err || echo "write rc=$?"
python3 tools/profile_summary.py $OUT $TAG
