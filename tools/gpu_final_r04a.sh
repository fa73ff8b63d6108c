#!/bin/bash
# GPU box, end of round 4, part A: counter files of the four driver-timed workloads (kernel trace + FETCH_SIZE + WRITE_SIZE passes,
# tools/gpu_profiles_r03.sh), SQ counters of configs 2 / 3 / 5, the driver's bench command.
set -o pipefail
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build_final.log 2>&1 || { echo BUILD FAILED; tail -5 gpurun_out/build_final.log; exit 1; }
bash tools/gpu_profiles_r03.sh r04_final
for c in 2 3 5; do
  if [ $c = 2 ]; then a="--config 2 --steps 200 --warmup 20"; elif [ $c = 3 ]; then a="--steps 200 --warmup 10"; else a="--config 5 --steps 20 --warmup 3"; fi
  bash tools/profile_sq.sh r04_final_c$c $a --no-cpu-baseline > gpurun_out/sq_c$c.txt 2>&1; echo "c$c sq done"
  cp gpurun_out/prof_r04_final_c$c/summary_sq_r04_final_c$c.txt gpurun_out/profiles_r04_final/rocprof_sq_counters_r04_final_c$c.txt
done
python bench.py --steps 20 --warmup 5 > gpurun_out/profiles_r04_final/bench_r04_driver_args.json 2>> gpurun_out/bench_final.err; tail -c 300 gpurun_out/bench_final.err
ls gpurun_out/profiles_r04_final
