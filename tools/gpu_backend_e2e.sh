#!/bin/bash
# GPU box: the drop-in binary end to end at BASELINE configs[2] size: wall time per observation step of the whole example_webmap
# run (2 172 observation steps, 17 381 control steps), in the three forms of the loop; -gpubusy adds per-launch event pairs
# (a few per cent of overhead), so every form is run with and without it.
B=slam_amd/bin/slam-backend
A="-m data/example_webmap.mat -method FASTSLAM2 -NPARTICLES 100000 -NEFFECTIVE 75000 -SWITCH_SEED_RANDOM 7"
for form in "-loop step" "" "-observe device"; do
  for busy in "" "-gpubusy 1"; do
    echo "== slam-backend $form $busy"
    $B $A $form $busy | grep -E "observation steps|GPU busy|mean loop"
  done
done
