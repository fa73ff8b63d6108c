#!/bin/bash
# GPU box: SQ counters of the update kernel (dynamic instruction mix / stall split), own PMC passes.
# usage: [KERNEL_FILTER=associate] tools/profile_sq.sh <tag> [bench args...]      (KERNEL_FILTER: substring of the kernel names to report; default update_kernel)
set -o pipefail
TAG=${1:-r02}; shift
ARGS=${@:---no-cpu-baseline}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 bench.py --single-pass --repeats 1 $ARGS > $OUT/bench_sq.json 2> $OUT/sq.err || echo "sq rc=$?"
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py --single-pass --repeats 1 $ARGS > $OUT/bench_sq2.json 2> $OUT/sq2.err || echo "sq2 rc=$?"
python3 - <<PY
import csv, glob, collections, json
steps = None
try:
    steps = json.loads(open("$OUT/bench_sq.json").read().strip().splitlines()[-1])["steps"]
except Exception:
    pass
rm = open("$OUT/summary_sq_$TAG.txt", "w")
for d in ("pmc_sq", "pmc_sq2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        per = collections.defaultdict(lambda: collections.defaultdict(dict))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:70]
            dsp = int(r["Dispatch_Id"])
            per[k][r["Counter_Name"]][dsp] = per[k][r["Counter_Name"]].get(dsp, 0.0) + float(r["Counter_Value"])
        for k, cs in per.items():
            if "${KERNEL_FILTER:-update_kernel}" not in k:
                continue
            parts = []
            for c, dv in sorted(cs.items()):
                vals = [dv[i] for i in sorted(dv)]
                if steps:
                    vals = vals[-steps:]
                parts.append("%s=%.5g" % (c, sum(vals) / max(len(vals), 1)))
            line = "%s | per dispatch, last %s dispatches | %s" % (k, steps, "  ".join(parts))
            print(line); rm.write(line + "\n")
PY
