#!/bin/bash
# GPU box, round 4 probe: new parity tests + where config 2's step goes (level stamps, SQ counters) + config 5 against the particle count
set -o pipefail
python -m pytest tests/test_gpu_resample_kat.py tests/test_gpu_observe.py -q -m gpu -s > gpurun_out/r4_kat.log 2>&1; grep -v "^$" gpurun_out/r4_kat.log | grep "resample KAT\|passed\|failed\|^FAILED" | tail -40
python -m pytest tests/test_gpu_freerun.py -q -m gpu -s -k "loop902" > gpurun_out/r4_free902.log 2>&1; grep "passed\|failed\|^FAILED\|^E  " gpurun_out/r4_free902.log | tail
SLAM_FREERUN_MEASURE=1 python -m pytest tests/test_gpu_freerun.py -q -m gpu -s -k "config5" > gpurun_out/r4_free10k.log 2>&1; grep "free-running\|passed\|failed\|^FAILED\|^E  " gpurun_out/r4_free10k.log | cut -c1-1800 | tail
make -s -C slam_amd/csrc stamps > gpurun_out/stamps_build.log 2>&1 || echo "stamps build failed"
timeout -k 10 300 python tools/stamps.py 1000 200 device FASTSLAM1 > gpurun_out/levels_c2_device.txt 2> gpurun_out/levels.err; echo "stamps rc=$?"
timeout -k 10 300 python tools/stamps.py 1000 200 host FASTSLAM1 > gpurun_out/levels_c2_host.txt 2>> gpurun_out/levels.err; echo "stamps rc=$?"
bash tools/profile_sq.sh r04_c2 --config 2 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/sq_c2.txt 2>&1; cat gpurun_out/sq_c2.txt | tail -4
for n in 50000 100000 200000; do
  python bench.py --config 5 --particles $n --steps 20 --warmup 3 --single-pass --repeats 1 --no-cpu-baseline > gpurun_out/bench_c5_N$n.json 2>> gpurun_out/bench_c5_n.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/bench_c5_N$n.json").read().strip().splitlines()[-1])
print("config 5, N=$n: %.4f ms/step, value %.4g, design GB/s %.0f" % (d["ms_per_step"], d["value"], d["roofline"]["design_GBps"]))
PY
done
