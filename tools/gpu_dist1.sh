#!/bin/bash
set -o pipefail
D=gpurun_out/d1; mkdir -p $D
python -c "import __graft_entry__ as g; g.build()" > $D/build.log 2>&1 || { echo BUILD FAILED; tail $D/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_dist.py -x -q --timeout 600 > $D/dist.log 2>&1; rc=$?; echo "dist rc=$rc"
tail -40 $D/dist.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 1100 python -m pytest tests -m gpu -q --timeout 900 --deselect tests/test_gpu_dist.py > $D/gputests.log 2>&1; echo "pytest rc=$?"
tail -15 $D/gputests.log
