#!/bin/bash
set -o pipefail
OUT=gpurun_out/prof_sq2; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --pmc SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_LDS SQ_WAVES --output-format csv -d $OUT/a -- python3 bench.py --steps 300 --warmup 100 --no-cpu-baseline > $OUT/a.json 2> $OUT/a.err || echo "rc=$?"
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/a/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(f)):
        a = agg[r["Kernel_Name"][:50]][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, cs in agg.items():
        print(k, " ".join("%s=%.4g" % (c, v[0] / max(v[1], 1)) for c, v in sorted(cs.items())))
PY
