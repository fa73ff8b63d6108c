#!/usr/bin/env python3
"""GPU box diagnostic (1 rank over RCCL): host-side time per phase of the sharded step."""
import os, sys, time
os.environ.setdefault("RANK", "0"); os.environ.setdefault("LOCAL_RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import numpy as np
import torch
import torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
import slam_amd
from slam_amd import host
from slam_amd.sharded import GpuEngine, ShardedFilter, TorchComm
n = 100096
tape = host.make_tape(["-m", os.path.join(ROOT, "data", "example_webmap.mat"), "-method", "FASTSLAM2", "-NPARTICLES", n, "-NEFFECTIVE", int(0.75 * n), "-SWITCH_SEED_RANDOM", 7], max_obs=900)
dev = torch.device("cuda", 0)
eng = GpuEngine(0, 1, n, tape["nlm"], method=2, n_effective=int(0.75 * n), rng_mode=slam_amd.RNG_PHILOX, seed=7, math_mode=1, device=0,
                external_stream=torch.cuda.current_stream().cuda_stream)
flt = ShardedFilter([eng], TorchComm(dev, stream_ordered=True), 1)
acc = {}
def timed(name, fn, *a):
    t = time.perf_counter(); r = fn(*a); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t; return r
E, c = flt.engines, flt.comm
ctl = [np.array(st["controls"], np.float32).reshape(-1, 3) for st in tape["steps"]]
def step(k, st):
    timed("step_local", E[0].step_local, ctl[k], tape["Q"], float(tape["dt"]), st["zf"], st["idf"], st["zn"], tape["R"], None, None)
    timed("totals", E[0].block_totals_into, c, flt.loc[0])
    timed("all_gather", c.all_gather, flt.loc, 2 * flt.nb_local, flt.gtot)
    plan = timed("plan(sync)", E[0].plan, c, flt.gtot[0], flt.nb_global)
    if plan.resampled:
        fields = E[0].record_floats()
        timed("ensure", flt._ensure, fields, [plan])
        timed("pack", E[0].pack, c, flt.gtot[0], flt.nb_global, plan, flt.send[0])
    timed("finish", E[0].finish, plan)
    timed("estimate_async", E[0].estimate_async)
for k, st in enumerate(tape["steps"][:100]):
    step(k, st)
torch.cuda.synchronize(); acc.clear()
t0 = time.perf_counter()
for k, st in enumerate(tape["steps"][100:900]):
    step(100 + k, st)
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print("total %.1f us/step" % (1e6 * tot / 800))
for k, v in acc.items():
    print("  %-16s %.1f us/step" % (k, 1e6 * v / 800))
flt.close()
dist.destroy_process_group()
