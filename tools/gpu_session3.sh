#!/bin/bash
# GPU box session 3: full GPU suite
set -o pipefail
D=gpurun_out/s3
mkdir -p $D
python -c "import __graft_entry__ as g; g.build()" > $D/build.log 2>&1 || { echo BUILD FAILED; tail -20 $D/build.log; exit 1; }
timeout -k 10 1100 python -m pytest tests -m gpu -q --timeout 900 > $D/gputests.log 2>&1; echo "pytest rc=$?"
tail -40 $D/gputests.log
