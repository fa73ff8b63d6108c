#!/bin/bash
set -o pipefail
D=gpurun_out/st5; mkdir -p $D
python -c "import __graft_entry__ as g; g.build()" > $D/build.log 2>&1 || { echo BUILD FAILED; exit 1; }
make -s -C slam_amd/csrc stamps > $D/stamps_build.log 2>&1 || { echo STAMPS BUILD FAILED; tail $D/stamps_build.log; exit 1; }
timeout -k 10 600 python tools/stamps_c5.py 12 > $D/stamps_c5.txt 2> $D/stamps_c5.err; echo "rc=$?"; tail -3 $D/stamps_c5.err; cat $D/stamps_c5.txt
