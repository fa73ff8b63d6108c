#!/bin/bash
# GPU box (round 6): the strict build three ways -- as shipped (the compiler's SLP vectoriser forms the
# v_pk_mul_f32 / v_pk_add_f32 it finds), without that pass (-fno-slp-vectorize: no packed FP32 at all), and with the 2x2 / 3x3
# algebra of the step paired BY HAND (-DSLAM_STRICT_PK=1) -- bit-identity of the three, config 3's step time, and the dynamic
# instruction counts per wave from the SQ counters.  Output: gpurun_out/strict_pk_r06/ (summary copied to profiles/ by hand).
# build first: make -C slam_amd/csrc svariant NAME=noslp EXTRA=-fno-slp-vectorize ; make -C slam_amd/csrc svariant NAME=pk EXTRA=-DSLAM_STRICT_PK=1
set -o pipefail
OUT=gpurun_out/strict_pk_r06
mkdir -p $OUT
for v in libslamgpu.so libslamgpu_snoslp.so libslamgpu_spk.so; do
  export SLAMGPU_LIB=$PWD/slam_amd/$v
  python3 tools/state_hash.py strict >> $OUT/hashes.txt 2>> $OUT/err.txt || exit 1
  python3 bench.py --math strict --single-pass --repeats 5 --no-cpu-baseline --steps 200 --warmup 20 > $OUT/bench_$v.json 2>> $OUT/err.txt || exit 1
  tools/profile_sq.sh strict_pk_$v --math strict --no-cpu-baseline --steps 200 --warmup 20 >> $OUT/sq.txt 2>> $OUT/err.txt || exit 1
done
python3 - <<'PY'
import json, glob, os
out = "gpurun_out/strict_pk_r06"
for f in sorted(glob.glob(out + "/bench_*.json")):
    j = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), "us per step %.2f" % (1e3 * j["ms_per_step"]), "window", j.get("window_repeats", {}).get("ms_per_step_min"), j.get("window_repeats", {}).get("ms_per_step_max"))
print(open(out + "/hashes.txt").read())
print(open(out + "/sq.txt").read())
PY
