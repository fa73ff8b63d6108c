#!/bin/bash
# rocprofv3 kernel-trace stats of the multi-GPU code path with one rank (both collectives), and of configs 2 and 4
set -o pipefail
D=gpurun_out/prof_dist; mkdir -p $D
python -c "import __graft_entry__ as g; g.build()" > $D/build.log 2>&1 || { echo BUILD FAILED; exit 1; }
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29781 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
for c in push native; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/$c -- python3 bench.py --gpus 1 --force-sharded --collective $c --no-check > $D/bench_$c.json 2> $D/$c.err || echo "$c rc=$?"
done
unset RANK WORLD_SIZE LOCAL_RANK
for cfg in 2 4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/c$cfg -- python3 bench.py --config $cfg --single-pass --no-cpu-baseline --no-strict > $D/bench_c$cfg.json 2> $D/c$cfg.err || echo "c$cfg rc=$?"
done
python3 - <<'PY'
import csv, glob, json
D="gpurun_out/prof_dist"
out=open(D+"/summary_dist.txt","w")
def p(s):
    print(s); out.write(s+"\n")
for tag, title in (("push","bench.py --gpus 1 --force-sharded --collective push  (world size 1)"),("native","bench.py --gpus 1 --force-sharded --collective native  (RCCL all-gather, world size 1)"),
                   ("c2","bench.py --config 2  (FASTSLAM1, 1 000 particles)"),("c4","bench.py --config 4  (FASTSLAM2, 1 001 472 particles in one context)")):
    fs=glob.glob("%s/%s/**/*kernel_stats.csv"%(D,tag), recursive=True)
    try:
        j=json.loads(open("%s/bench_%s.json"%(D,tag)).read().strip().splitlines()[-1]); head="value %.4g particle-updates/s, %.3f us per step" % (j["value"], 1e3*j["ms_per_step"])
    except Exception as e:
        head="(bench line unreadable: %s)"%e
    p("== %s: %s" % (title, head))
    for f in fs[:1]:
        for r in list(csv.DictReader(open(f)))[:6]:
            p("   %-110s calls %6s avg_ns %10.1f pct %6s" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]), r["Percentage"]))
PY
