#!/bin/bash
# which memory-side counters exist, and TLB / L2 / EA behaviour of the config-5 update kernel (own PMC passes)
set -o pipefail
D=gpurun_out/cnt5; mkdir -p $D
python -c "import __graft_entry__ as g; g.build()" > $D/build.log 2>&1 || { echo BUILD FAILED; exit 1; }
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --list-avail > $D/avail.txt 2>&1 || true
grep -c . $D/avail.txt
grep -o -E "\b(TCP|TCC|TCA|UTCL2|GRBM|TA|TD|SQC?)_[A-Z0-9_]*(UTCL|TLB|MISS|HIT|EA_|STALL|TAG|DRAM|BUSY|LATENCY)[A-Z0-9_]*" $D/avail.txt | sort -u > $D/names.txt
wc -l $D/names.txt
ARGS="--config 5 --no-strict --no-cpu-baseline --single-pass --steps 12 --warmup 3"
run() { tag=$1; shift; timeout -k 10 400 rocprofv3 --pmc "$@" --output-format csv -d $D/$tag -- python3 bench.py $ARGS > $D/$tag.json 2> $D/$tag.err || echo "$tag rc=$?"; }
run p1 TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_REQUEST TCP_TCC_READ_REQ
run p2 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
run p3 TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_RD_UNCACHED_32B_sum TCC_TAG_STALL_sum
run p4 TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_TCC_WRITE_REQ_LATENCY TCP_TA_TCP_STATE_READ
python3 - <<'PY'
import csv, glob, collections
D="gpurun_out/cnt5"
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
                v=[dv[i] for i in sorted(dv)][-12:]
                parts.append("%s=%.5g"%(c,sum(v)/len(v)))
            line="%s %s | %s"%(tag,k,"  ".join(parts)); print(line); out.write(line+"\n")
    for e in glob.glob("%s/%s.err"%(D,tag)):
        t=open(e).read()
        if "rror" in t: print(tag,"ERR:",t[-300:])
PY
