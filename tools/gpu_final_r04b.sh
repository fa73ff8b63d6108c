#!/bin/bash
# GPU box, end of round 4, part B: L2 / fabric counters of config 5 (tools/profile_tcc.sh), level stamps (config 2 host / device
# front end, config 3), one shard's launch against the width of the gathered totals table (tools/dist_width.py, twice: the
# logical-shard timing is noisy), the 2 000-step windows, the drop-in binary end to end.
set -o pipefail
mkdir -p gpurun_out/profiles_r04_final
make -s -C slam_amd/csrc stamps > gpurun_out/stamps_build.log 2>&1 || echo "stamps build failed"
bash tools/profile_tcc.sh r04_c5 > gpurun_out/tcc_c5.txt 2>&1; tail -12 gpurun_out/tcc_c5.txt
cp gpurun_out/prof_tcc_r04_c5/summary_tcc_r04_c5.txt gpurun_out/profiles_r04_final/rocprof_tcc_counters_r04_c5.txt
timeout -k 10 300 python tools/stamps.py 1000 200 host FASTSLAM1 > gpurun_out/profiles_r04_final/update_kernel_levels_r04_config2.txt 2> gpurun_out/levels.err; echo "stamps rc=$?"
timeout -k 10 300 python tools/stamps.py 1000 200 device FASTSLAM1 > gpurun_out/profiles_r04_final/update_kernel_levels_r04_config2_device_front_end.txt 2>> gpurun_out/levels.err; echo "stamps rc=$?"
timeout -k 10 400 python tools/stamps.py 100000 200 > gpurun_out/profiles_r04_final/update_kernel_levels_r04_N100000.txt 2>> gpurun_out/levels.err; echo "stamps rc=$?"
for r in 1 2; do python tools/dist_width.py; done > gpurun_out/profiles_r04_final/dist_width_r04.txt 2>&1; cat gpurun_out/profiles_r04_final/dist_width_r04.txt
for o in host device; do
  python bench.py --steps 2000 --warmup 100 --single-pass --repeats 3 --no-cpu-baseline --observe $o > gpurun_out/profiles_r04_final/bench_r04_c3_2000steps_observe_$o.json 2>> gpurun_out/bench_final.err
done
bash tools/gpu_backend_e2e.sh > gpurun_out/profiles_r04_final/slam_backend_e2e_r04.txt 2>&1; echo "e2e rc=$?"
ls gpurun_out/profiles_r04_final
