#!/bin/bash
# GPU box: SQ counters of the per-particle kernels (dynamic instruction mix / stall split), own PMC pass.
set -o pipefail
TAG=${1:-r01}; shift
ARGS=${@:---steps 300 --warmup 100 --no-cpu-baseline}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 bench.py $ARGS > $OUT/bench_sq.json 2> $OUT/sq.err || echo "sq rc=$?"
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py $ARGS > $OUT/bench_sq2.json 2> $OUT/sq2.err || echo "sq2 rc=$?"
python3 - <<PY
import csv, glob, collections
for d in ("pmc_sq", "pmc_sq2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            a = agg[k][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
        with open("$OUT/summary_sq_$TAG.txt", "a") as o:
            for k, cs in agg.items():
                line = k + " | " + "  ".join("%s=%.4g" % (c, v[0] / max(v[1], 1)) for c, v in sorted(cs.items()))
                print(line); o.write(line + "\n")
PY
