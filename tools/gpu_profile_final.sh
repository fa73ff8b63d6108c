#!/bin/bash
# final profiles of the round: kernel-trace stats + FETCH/WRITE PMC passes + SQ counters for configs 3 and 5, level stamps
set -o pipefail
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build_final.log 2>&1 || { echo BUILD FAILED; exit 1; }
make -s -C slam_amd/csrc stamps > gpurun_out/stamps_build.log 2>&1 || echo "stamps build failed"
bash tools/profile.sh r02_c3_final --config 3 --no-cpu-baseline --no-strict; echo "c3 profile done"
bash tools/profile.sh r02_c5_final --config 5 --no-cpu-baseline --no-strict; echo "c5 profile done"
bash tools/profile_sq.sh r02_c3_final --config 3 --no-cpu-baseline --no-strict > gpurun_out/sq_c3.txt 2>&1; echo "c3 sq done"
bash tools/profile_sq.sh r02_c5_final --config 5 --no-cpu-baseline --no-strict > gpurun_out/sq_c5.txt 2>&1; echo "c5 sq done"
timeout -k 10 600 python tools/stamps.py 100000 200 > gpurun_out/levels_c3_final.txt 2> gpurun_out/levels_c3_final.err; echo "stamps c3 rc=$?"
timeout -k 10 600 python tools/stamps_c5.py 12 > gpurun_out/levels_c5_final.txt 2> gpurun_out/levels_c5_final.err; echo "stamps c5 rc=$?"
ls gpurun_out/prof_r02_c3_final gpurun_out/prof_r02_c5_final
