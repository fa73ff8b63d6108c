#!/bin/bash
# GPU box: counter files of the four driver-timed workloads of bench.py's default line (tools/profile.sh each: kernel trace +
# FETCH_SIZE pass + WRITE_SIZE pass), copied into profiles/ under the tag given (e.g. r03_before, r03_final).
# usage: tools/gpu_profiles_r03.sh <suffix>
set -o pipefail
S=${1:-r03}
run() {  # tag, bench args
    local tag=$1; shift
    bash tools/profile.sh $tag "$@" > gpurun_out/prof_$tag.log 2>&1 || echo "profile $tag rc=$?"
    cp gpurun_out/prof_$tag/traffic_$tag.json profiles/traffic_$tag.json
    cp gpurun_out/prof_$tag/summary_$tag.txt profiles/rocprof_summary_$tag.txt
    f=$(find gpurun_out/prof_$tag/trace -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f profiles/rocprof_kernel_stats_$tag.csv
    grep "timed window" gpurun_out/prof_$tag/summary_$tag.txt
}
run ${S}_c3 --steps 20 --warmup 5 --no-cpu-baseline
run ${S}_c3_strict --math strict --steps 20 --warmup 5 --no-cpu-baseline
run ${S}_c2 --config 2 --steps 200 --warmup 20 --no-cpu-baseline
run ${S}_c5 --config 5 --steps 20 --warmup 3 --no-cpu-baseline
mkdir -p gpurun_out/profiles_$S && cp profiles/*${S}_c* gpurun_out/profiles_$S/
