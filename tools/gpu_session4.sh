#!/bin/bash
# GPU box session 4: full suite again + level stamps + rocprof of config 3 and config 5
set -o pipefail
D=gpurun_out/s4
mkdir -p $D
python -c "import __graft_entry__ as g; g.build()" > $D/build.log 2>&1 || { echo BUILD FAILED; tail -20 $D/build.log; exit 1; }
make -s -C slam_amd/csrc stamps >> $D/build.log 2>&1
timeout -k 10 1100 python -m pytest tests -m gpu -q --timeout 900 > $D/gputests.log 2>&1; echo "pytest rc=$?"
tail -8 $D/gputests.log
timeout -k 10 300 python tools/stamps.py 100000 200 > $D/stamps_N100000.txt 2>&1; echo "stamps rc=$?"; cat $D/stamps_N100000.txt
timeout -k 10 300 python tools/stamps.py 1024 100 > $D/stamps_N1024.txt 2>&1; echo "stamps1024 rc=$?"; cat $D/stamps_N1024.txt
timeout -k 10 600 bash tools/profile.sh r02_c3 --no-cpu-baseline > $D/profile_c3.log 2>&1; echo "prof c3 rc=$?"; tail -30 $D/profile_c3.log
timeout -k 10 900 bash tools/profile.sh r02_c5 --config 5 --no-cpu-baseline > $D/profile_c5.log 2>&1; echo "prof c5 rc=$?"; tail -30 $D/profile_c5.log
