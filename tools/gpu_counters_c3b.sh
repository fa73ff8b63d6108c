#!/bin/bash
# instruction-cache / scalar-cache / per-type latency counters of the config-3 update kernel (own PMC passes)
set -o pipefail
D=gpurun_out/cnt3b; mkdir -p $D
python -c "import __graft_entry__ as g; g.build()" > $D/build.log 2>&1 || { echo BUILD FAILED; exit 1; }
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
ARGS="--config 3 --no-strict --no-cpu-baseline --single-pass --steps 400 --warmup 20"
run() { tag=$1; shift; timeout -k 10 400 rocprofv3 --pmc "$@" --output-format csv -d $D/$tag -- python3 bench.py $ARGS > $D/$tag.json 2> $D/$tag.err || echo "$tag rc=$?"; }
run q1 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
run q2 SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE
run q3 SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM
run q4 SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS
run q5 SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES
python3 - <<'PY'
import csv, glob, collections
D="gpurun_out/cnt3b"
out=open(D+"/summary.txt","w")
for tag in ("q1","q2","q3","q4","q5"):
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
    for e in glob.glob("%s/%s.err"%(D,tag)):
        t=open(e).read()
        if "rror" in t: print(tag,"ERR:",t[-300:])
PY
