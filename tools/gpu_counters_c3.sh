#!/bin/bash
# TLB / L2 / latency counters of the config-3 update kernel (own PMC passes)
set -o pipefail
D=gpurun_out/cnt3; mkdir -p $D
python -c "import __graft_entry__ as g; g.build()" > $D/build.log 2>&1 || { echo BUILD FAILED; exit 1; }
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
ARGS="--config 3 --no-strict --no-cpu-baseline --single-pass --steps 400 --warmup 20"
run() { tag=$1; shift; timeout -k 10 400 rocprofv3 --pmc "$@" --output-format csv -d $D/$tag -- python3 bench.py $ARGS > $D/$tag.json 2> $D/$tag.err || echo "$tag rc=$?"; }
run p1 TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_REQUEST TCP_TCC_READ_REQ
run p2 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
run p3 TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum TCC_REQ_sum TCC_TAG_STALL_sum
run p4 TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_TCC_WRITE_REQ_LATENCY TCP_TCC_WRITE_REQ
python3 - <<'PY'
import csv, glob, collections
D="gpurun_out/cnt3"
out=open(D+"/summary.txt","w")
for tag in ("p1","p2","p3","p4"):
    for f in glob.glob("%s/%s/**/*counter_collection.csv"%(D,tag), recursive=True):
        per=collections.defaultdict(lambda: collections.defaultdict(dict))
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0][:60]; d=int(r["Dispatch_Id"])
            per[k][r["Counter_Name"]][d]=per[k][r["Counter_Name"]].get(d,0.0)+float(r["Counter_Value"])
        for k,cs in per.items():
            if "update_kernel" not in k: continue
            parts=[]
            for c,dv in sorted(cs.items()):
                v=[dv[i] for i in sorted(dv)][-400:]
                parts.append("%s=%.5g"%(c,sum(v)/len(v)))
            line="%s %s | %s"%(tag,k,"  ".join(parts)); print(line); out.write(line+"\n")
PY
