#!/bin/bash
set -o pipefail
D=gpurun_out/s9; mkdir -p $D
python -c "import __graft_entry__ as g; g.build()" > $D/build.log 2>&1 || { echo BUILD FAILED; tail $D/build.log; exit 1; }
timeout -k 10 1100 python -m pytest tests -m gpu -q --timeout 900 > $D/gputests.log 2>&1; echo "pytest rc=$?"
tail -30 $D/gputests.log
