#!/bin/bash
# 1-rank rehearsal of the multi-GPU bench path (RCCL all-gather at world size 1) beside the single-context bench
set -o pipefail
D=gpurun_out/d2; mkdir -p $D
python -c "import __graft_entry__ as g; g.build()" > $D/build.log 2>&1 || { echo BUILD FAILED; tail $D/build.log; exit 1; }
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29711 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
timeout -k 10 600 python bench.py --gpus 1 --force-sharded --steps 2000 --warmup 100 > $D/dist_n1.json 2> $D/dist_n1.err; echo "dist rc=$?"
tail -3 $D/dist_n1.err; cat $D/dist_n1.json
timeout -k 10 600 python bench.py --gpus 1 --force-sharded --mgpu exchange --steps 2000 --warmup 100 > $D/exch_n1.json 2> $D/exch_n1.err; echo "exchange rc=$?"
tail -3 $D/exch_n1.err; cat $D/exch_n1.json
unset RANK WORLD_SIZE LOCAL_RANK
timeout -k 10 600 python bench.py --no-cpu-baseline --no-strict > $D/single.json 2> $D/single.err; echo "single rc=$?"
python - <<'PY'
import json
for f in ("dist_n1","exch_n1","single"):
    try:
        j=json.loads(open("gpurun_out/d2/%s.json"%f).read().strip().splitlines()[-1]); print(f, j["value"], j["ms_per_step"], j["config"].get("check_vs_single_context"))
    except Exception as e: print(f, "ERR", e)
PY
