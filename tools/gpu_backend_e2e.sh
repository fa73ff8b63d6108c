#!/bin/bash
# GPU box: the drop-in binary end to end at BASELINE configs[2] size: wall time per observation step of the whole example_webmap
# run (2 172 observation steps, 17 381 control steps), in the three forms of the loop; -gpubusy adds per-launch event pairs
# (a few per cent of overhead), so every form is run with and without it.
B=slam_amd/bin/slam-backend
A="-m data/example_webmap.mat -method FASTSLAM2 -NPARTICLES 100000 -NEFFECTIVE 75000 -SWITCH_SEED_RANDOM 7"
for form in "-loop step" "" "-observe device"; do
  for busy in "" "-gpubusy 1"; do
    echo "== slam-backend $form $busy"
    $B $A $form $busy | grep -E "observation steps|GPU busy|mean loop|host side"
  done
done
# round 5: BASELINE config 2's size through the binary: -observe device hands the loop over 256 iterations at a time, which small
# compact contexts run as ONE launch each (the persistent step loop); SLAMGPU_NO_PERSIST=1: the same calls as loops of launches
A2="-m data/example_webmap.mat -method FASTSLAM1 -NPARTICLES 1000 -NEFFECTIVE 750 -SWITCH_SEED_RANDOM 7"
for env in "" "SLAMGPU_NO_PERSIST=1"; do
  echo "== $env slam-backend FASTSLAM1 1000 particles -observe device"
  env $env $B $A2 -observe device | grep -E "observation steps|GPU busy|mean loop|host side"
done
echo "== slam-backend FASTSLAM1 1000 particles (host front end, one slamgpu_step per observation)"
$B $A2 | grep -E "observation steps|GPU busy|mean loop|host side"
