for m in example_loop1 example_loop2; do timeout -k 10 300 python tools/particle_assoc_probe.py --seeds 7-9 --particles 512,2048 --map $m --tag $m,fs2 --max-landmarks 999; done
timeout -k 10 500 python tools/particle_assoc_probe.py --seeds 7-8 --particles 2048 --map example_loop902 --tag loop902,fs2 --max-landmarks 999
