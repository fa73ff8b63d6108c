#!/bin/bash
# GPU box: kernel durations of one shard's launch against the table width, G logical shards on one GPU, under rocprofv3
# --kernel-trace --stats (one run per G and library: usage tools/gpu_dist_width_prof.sh [other library name])
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OTHER=$1
for G in 4 8; do
  for v in this $OTHER; do
    if [ $v = this ]; then unset SLAMGPU_LIB; else export SLAMGPU_LIB=$PWD/slam_amd/libslamgpu_$v.so; fi
    D=gpurun_out/dwp_${G}_$v; rm -rf $D
    rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 tools/dist_width.py 100096 $G > $D.log 2>&1
    python3 - <<PY
import csv, glob
for f in glob.glob("$D/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "update_kernel<2, 2, false>" in r["Name"]:
            print("G=$G [%-5s] update_kernel<2,2,false>: %s calls, average %.2f us (min %.2f, max %.2f)" % ("$v", r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
  done
done
