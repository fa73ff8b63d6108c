#!/bin/bash
# GPU box: L2 / fabric-side counters of the update kernel (one small PMC pass each: the TCC has few slots), for the floor
# analysis of BASELINE config 5: hit / miss, requests to DRAM, credit stalls (the memory side pushing back), queue levels.
# usage: tools/profile_tcc.sh <tag> [bench args...]
set -o pipefail
TAG=${1:-r04_c5}; shift
ARGS=${@:---config 5 --steps 20 --warmup 3 --no-cpu-baseline}
OUT=gpurun_out/prof_tcc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
i=0
for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
           "TCC_BUSY_sum TCC_CYCLE_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum" "TCC_TAG_STALL_sum TCC_REQ_sum" \
           "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum" "TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 bench.py --single-pass --repeats 1 $ARGS > $OUT/bench_$i.json 2> $OUT/p$i.err || echo "pass $i rc=$?"
done
python3 - <<PY
import csv, glob, collections, json
steps = 20
try:
    steps = json.loads(open("$OUT/bench_1.json").read().strip().splitlines()[-1])["steps"]
except Exception:
    pass
rm = open("$OUT/summary_tcc_$TAG.txt", "w")
tot = {}
for f in sorted(glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(dict))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:70]
        dsp = int(r["Dispatch_Id"])
        per[k][r["Counter_Name"]][dsp] = per[k][r["Counter_Name"]].get(dsp, 0.0) + float(r["Counter_Value"])
    for k, cs in per.items():
        if "update_kernel" not in k:
            continue
        for c, dv in sorted(cs.items()):
            vals = [dv[i] for i in sorted(dv)][-steps:]
            tot[c] = sum(vals) / max(len(vals), 1)
for c in sorted(tot):
    line = "%-44s %.6g per dispatch (mean of the last %d dispatches of update_kernel)" % (c, tot[c], steps)
    print(line); rm.write(line + "\n")
def ratio(a, b, label):
    if a in tot and b in tot and tot[b]:
        line = "%-44s %.4f" % (label, tot[a] / tot[b])
        print(line); rm.write(line + "\n")
if "TCC_HIT_sum" in tot and "TCC_MISS_sum" in tot:
    line = "%-44s %.4f" % ("L2 hit rate HIT / (HIT + MISS)", tot["TCC_HIT_sum"] / (tot["TCC_HIT_sum"] + tot["TCC_MISS_sum"]))
    print(line); rm.write(line + "\n")
ratio("TCC_BUSY_sum", "TCC_CYCLE_sum", "TCC busy / cycle")
ratio("TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum", "TCC_CYCLE_sum", "read DRAM-credit stall / TCC cycle")
ratio("TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum", "TCC_CYCLE_sum", "write DRAM-credit stall / TCC cycle")
ratio("TCC_EA0_WRREQ_STALL_sum", "TCC_CYCLE_sum", "write-request stall / TCC cycle")
ratio("TCC_EA0_RDREQ_LEVEL_sum", "TCC_EA0_RDREQ_sum", "mean read latency at the fabric (LEVEL / REQ, TCC cycles)")
ratio("TCC_EA0_WRREQ_LEVEL_sum", "TCC_EA0_WRREQ_sum", "mean write latency at the fabric (LEVEL / REQ, TCC cycles)")
ratio("TCC_TAG_STALL_sum", "TCC_CYCLE_sum", "tag stall / TCC cycle")
PY
