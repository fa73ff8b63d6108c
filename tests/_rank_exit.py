"""helper of test_bench_launch_cpu.py: rank 1 exits with the code given on the command line, rank 0 would wait forever"""
import os
import sys
import time

if os.environ["RANK"] == "1":
    sys.exit(int(sys.argv[1]))
time.sleep(120)
