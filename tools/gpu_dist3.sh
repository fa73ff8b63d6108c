#!/bin/bash
set -o pipefail
D=gpurun_out/d3; mkdir -p $D
python -c "import __graft_entry__ as g; g.build()" > $D/build.log 2>&1 || { echo BUILD FAILED; tail $D/build.log; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_backend_cli.py -x -q --timeout 600 > $D/dist.log 2>&1; rc=$?; echo "dist rc=$rc"
tail -40 $D/dist.log
[ $rc -eq 0 ] || exit $rc
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29711 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
for coll in native torch; do
timeout -k 10 600 python bench.py --gpus 1 --force-sharded --collective $coll --steps 2000 --warmup 100 > $D/dist_$coll.json 2> $D/dist_$coll.err; echo "$coll rc=$?"
tail -2 $D/dist_$coll.err
done
unset RANK WORLD_SIZE LOCAL_RANK
timeout -k 10 600 python bench.py --no-cpu-baseline --no-strict > $D/single.json 2> $D/single.err; echo "single rc=$?"
python - <<'PY'
import json
for f in ("dist_native","dist_torch","single"):
    try:
        j=json.loads(open("gpurun_out/d3/%s.json"%f).read().strip().splitlines()[-1]); print(f, j["value"], j["ms_per_step"], j["config"].get("check_vs_single_context"))
    except Exception as e: print(f, "ERR", e)
PY
