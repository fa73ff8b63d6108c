#!/bin/bash
set -o pipefail
D=gpurun_out/s7; mkdir -p $D
python -c "import __graft_entry__ as g; g.build()" > $D/build.log 2>&1 || { echo BUILD FAILED; exit 1; }
timeout -k 10 900 bash tools/profile.sh r02_c5 --config 5 --no-cpu-baseline > $D/profile_c5.log 2>&1; echo "prof c5 rc=$?"; grep -E "timed window|update_kernel" $D/profile_c5.log | head
timeout -k 10 900 bash tools/profile_sq.sh r02_c5 --config 5 --no-cpu-baseline > $D/profile_sq_c5.log 2>&1; echo "sq c5 rc=$?"; tail -4 $D/profile_sq_c5.log
timeout -k 10 600 bash tools/profile.sh r02_c3 --no-cpu-baseline > $D/profile_c3.log 2>&1; echo "prof c3 rc=$?"; grep -E "timed window|update_kernel" $D/profile_c3.log | head
timeout -k 10 600 bash tools/profile_sq.sh r02_c3 --no-cpu-baseline > $D/profile_sq_c3.log 2>&1; echo "sq c3 rc=$?"; tail -4 $D/profile_sq_c3.log
