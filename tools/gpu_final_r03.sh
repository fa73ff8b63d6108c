#!/bin/bash
# GPU box, end of round 3: counter files of the four driver-timed workloads (tools/gpu_profiles_r03.sh) and of config 3 with
# the observation made inside the launch, SQ counters of configs 3 and 5, the level stamps (host / device front end), the
# drop-in binary end to end, the driver's bench command.
set -o pipefail
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build_final.log 2>&1 || { echo BUILD FAILED; exit 1; }
make -s -C slam_amd/csrc stamps > gpurun_out/stamps_build.log 2>&1 || echo "stamps build failed"
bash tools/gpu_profiles_r03.sh r03_final
bash tools/profile.sh r03_final_c3_device --observe device --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prof_r03_final_c3_device.log 2>&1 || echo "profile c3 device rc=$?"
cp gpurun_out/prof_r03_final_c3_device/traffic_r03_final_c3_device.json gpurun_out/prof_r03_final_c3_device/summary_r03_final_c3_device.txt gpurun_out/profiles_r03_final/ 2>/dev/null
bash tools/profile_sq.sh r03_final_c3 --steps 200 --warmup 10 --no-cpu-baseline > gpurun_out/sq_c3.txt 2>&1; echo "c3 sq done"
bash tools/profile_sq.sh r03_final_c5 --config 5 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/sq_c5.txt 2>&1; echo "c5 sq done"
cp gpurun_out/prof_r03_final_c3/summary_sq_r03_final_c3.txt gpurun_out/profiles_r03_final/rocprof_sq_counters_r03_final_c3.txt
cp gpurun_out/prof_r03_final_c5/summary_sq_r03_final_c5.txt gpurun_out/profiles_r03_final/rocprof_sq_counters_r03_final_c5.txt
timeout -k 10 400 python tools/stamps.py 100000 200 > gpurun_out/profiles_r03_final/update_kernel_levels_r03_N100000.txt 2> gpurun_out/levels.err; echo "stamps rc=$?"
timeout -k 10 400 python tools/stamps.py 100000 200 device > gpurun_out/profiles_r03_final/update_kernel_levels_r03_N100000_device_front_end.txt 2>> gpurun_out/levels.err; echo "stamps rc=$?"
timeout -k 10 300 python tools/stamps.py 1024 200 > gpurun_out/profiles_r03_final/update_kernel_levels_r03_N1024.txt 2>> gpurun_out/levels.err; echo "stamps rc=$?"
cp profiles/*r03_final* gpurun_out/profiles_r03_final/ 2>/dev/null
bash tools/gpu_backend_e2e.sh > gpurun_out/profiles_r03_final/slam_backend_e2e_r03.txt 2>&1; echo "e2e rc=$?"
for o in host device; do
  python bench.py --steps 2000 --warmup 100 --single-pass --no-cpu-baseline --observe $o > gpurun_out/profiles_r03_final/bench_r03_c3_2000steps_observe_$o.json 2>> gpurun_out/bench_final.err
done
python bench.py --config 5 --observe device --steps 20 --warmup 3 --single-pass --no-cpu-baseline > gpurun_out/profiles_r03_final/bench_r03_c5_observe_device.json 2>> gpurun_out/bench_final.err
python bench.py --steps 20 --warmup 5 > gpurun_out/profiles_r03_final/bench_r03_driver_args.json 2>> gpurun_out/bench_final.err; tail -c 300 gpurun_out/bench_final.err
ls gpurun_out/profiles_r03_final
