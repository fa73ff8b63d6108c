#!/bin/bash
# GPU box (round 4): BASELINE config 5 against the particle count -- where the launch saturates and what the 391-block
# geometry costs (256 CUs: 135 hold two blocks, 121 one) -- 50 176 / 65 536 (one block per CU) / 100 000 / 131 072 (two blocks
# on every CU) / 200 000 particles
for n in 50176 65536 100000 131072 200000; do
  python bench.py --config 5 --particles $n --steps 20 --warmup 3 --single-pass --repeats 3 --no-cpu-baseline > gpurun_out/bench_c5_N$n.json 2>> gpurun_out/bench_c5_n.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/bench_c5_N$n.json").read().strip().splitlines()[-1])
print("config 5, N=%7d: %.4f ms per step (repeats %s), %.4g particle-updates/s, %.1f ns per particle and step, design bytes %.0f GB/s" % (
    $n, d["ms_per_step"], [round(x, 4) for x in d["window_repeats"]["ms_per_step_all"]], d["value"], 1e6 * d["ms_per_step"] / $n, d["roofline"]["design_GBps"]))
PY
done
