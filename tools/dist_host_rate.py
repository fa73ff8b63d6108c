#!/usr/bin/env python3
"""GPU box: host cost of one distributed filter step (slamgpu_dist_step: update launch + flag barrier launch, or + RCCL
all-gather) against its device time, world size 1.  The host must stay ahead of the GPU for the step loop to be GPU-bound.
usage: MASTER_ADDR=127.0.0.1 MASTER_PORT=29771 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python tools/dist_host_rate.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402  (before libslamgpu: one HIP runtime)
import torch.distributed as dist  # noqa: E402

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
import slam_amd as sg  # noqa: E402
from slam_amd import host  # noqa: E402
from slam_amd.dist import DistFilter, NativeGather  # noqa: E402

n, start, steps = 100096, 600, 1500
dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
tp = host.make_tape(["-m", os.path.join(ROOT, "data", "example_webmap.mat"), "-method", "FASTSLAM2", "-NPARTICLES", n, "-NEFFECTIVE",
                     int(0.75 * n), "-SWITCH_SEED_RANDOM", 7], max_obs=start + steps)
for push in (False, True):
    ctx = sg.SlamGpu(n, tp["nlm"], method=sg.FASTSLAM2, n_effective=int(0.75 * n), rng_mode=sg.RNG_PHILOX, seed=7, math_mode=1,
                     external_stream=torch.cuda.current_stream().cuda_stream)
    f = DistFilter([ctx], NativeGather(ctx, dev))
    if push:
        assert f.use_push()
    calls = [f.prepare_step(np.array(st["controls"], np.float32).reshape(-1, 3), tp["Q"], float(tp["dt"]), st["zf"], st["idf"], st["zn"], tp["R"])
             for st in tp["steps"]]
    for c in calls[:start]:
        c()
    f.history_fetch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for c in calls[start:start + steps]:
        c()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("%s: host enqueue %.2f us per step, completion %.2f us per step" % ("push + flag barrier" if push else "RCCL all-gather", 1e6 * t_host / steps, 1e6 * t_all / steps))
    f.history_fetch()
    f.close()
dist.destroy_process_group()
