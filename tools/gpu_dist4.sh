#!/bin/bash
# 1-rank rehearsal of the driver's multi-GPU launch line (torch.distributed.run) + the plain single-GPU line
set -o pipefail
D=gpurun_out/d4; mkdir -p $D
python -c "import __graft_entry__ as g; g.build()" > $D/build.log 2>&1 || { echo BUILD FAILED; exit 1; }
timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29733 bench.py --gpus 1 --force-sharded --steps 2000 --warmup 100 > $D/dist1.json 2> $D/dist1.err; echo "dist rc=$?"
tail -2 $D/dist1.err; tail -1 $D/dist1.json | cut -c1-400
timeout -k 10 900 python bench.py > $D/single.json 2> $D/single.err; echo "single rc=$?"; tail -1 $D/single.json | cut -c1-300
python - <<'PY'
import json
for f in ("dist1","single"):
    j=json.loads(open("gpurun_out/d4/%s.json"%f).read().strip().splitlines()[-1]); print(f, "%.4g"%j["value"], "%.5f"%j["ms_per_step"], j["config"].get("check_vs_single_context"), j["config"].get("multi_gpu_path"))
PY
