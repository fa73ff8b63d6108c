#!/bin/bash
# how many landmarks does the gated-association run end with? (fast / strict builds, a few seeds; this tree and round 4's)
for exe in slam_amd/bin/slam-backend tools/scratch/r4/slam_amd/bin/slam-backend; do for seed in 7 8; do for math in fast strict; do
  n=$($exe -m data/example_webmap.mat -method FASTSLAM2 -NPARTICLES 512 -NEFFECTIVE 384 -SWITCH_SEED_RANDOM $seed -assoc gated -math $math 2>&1 | grep -o "landmarks in map: [0-9]*")
  echo "$exe seed $seed $math: $n"
done; done; done
