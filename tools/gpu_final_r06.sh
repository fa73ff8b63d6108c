#!/bin/bash
# GPU box, end of round 6: counter files of the driver-timed workloads of bench.py's default line (kernel trace + FETCH_SIZE +
# WRITE_SIZE passes each: tools/profile.sh), SQ counters of config 2's persistent loop, the driver's bench command.
# usage: bash tools/gpu_final_r06.sh [suffix]
set -o pipefail
S=${1:-r06_final}
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build_final.log 2>&1 || { echo BUILD FAILED; tail -5 gpurun_out/build_final.log; exit 1; }
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
run ${S}_c2 --config 2 --steps 200 --warmup 20 --observe batched --no-cpu-baseline
run ${S}_c5 --config 5 --steps 20 --warmup 3 --no-cpu-baseline
run ${S}_c6 --config 6 --steps 20 --warmup 5 --no-cpu-baseline
run ${S}_c4 --config 4 --steps 20 --warmup 5 --no-cpu-baseline
mkdir -p gpurun_out/profiles_$S && cp profiles/*${S}_c* gpurun_out/profiles_$S/
# (raw counter dumps are tens of MB per workload: only the condensed files go back)
for t in c3 c3_strict c2 c5 c6 c4; do rm -rf gpurun_out/prof_${S}_$t/trace gpurun_out/prof_${S}_$t/pmc_fetch gpurun_out/prof_${S}_$t/pmc_write; done
python bench.py --steps 20 --warmup 5 > gpurun_out/profiles_$S/bench_r06_driver_args.json 2>> gpurun_out/bench_final.err; tail -c 300 gpurun_out/bench_final.err
# (config 4 with its default window of 2 000 steps: a 20-step window at 10^6 particles ends before the part's sustained state)
python bench.py --config 4 --no-also --no-cpu-baseline > gpurun_out/profiles_$S/bench_r06_c4.json 2>> gpurun_out/bench_final.err
ls gpurun_out/profiles_$S
