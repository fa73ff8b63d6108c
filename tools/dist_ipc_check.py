"""Two processes, one shard each, both on GPU 0: the distributed contexts map each other's state through hipIpc and must
reproduce the single-context run bit for bit.  The all-gather of the block totals goes through torch.distributed/gloo on
host copies (RCCL refuses two ranks on one device; on a multi-GPU node the same loop runs with TorchGather = RCCL).
Run by tests/test_gpu_dist.py; prints DIST_IPC_OK."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class GlooGather:
    """totals through the host: device -> numpy -> gloo all_gather -> device (test plumbing only)"""

    def __init__(self, ctx, rank, world):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.ctx, self.rank, self.world = torch, dist, [ctx], rank, world
        self.shards = [rank]
        self.host = None

    def exchange_blobs(self, blobs):
        out = [None] * self.world
        self.dist.all_gather_object(out, blobs[0])
        return out

    def all_gather(self):
        import ctypes as C
        c = self.ctx[0]
        loc, gat, n = c.dist_totals()
        c.sync()
        hip = C.CDLL("libamdhip64.so")
        mine = np.zeros(n, np.float32)
        assert hip.hipMemcpy(mine.ctypes.data_as(C.c_void_p), C.c_void_p(loc), C.c_size_t(4 * n), 2) == 0
        t = self.torch.from_numpy(mine)
        out = [self.torch.zeros(n) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        allv = np.concatenate([o.numpy() for o in out]).astype(np.float32)
        assert hip.hipMemcpy(C.c_void_p(gat), allv.ctypes.data_as(C.c_void_p), C.c_size_t(4 * n * self.world), 1) == 0

    def all_gather_rows(self, rows):
        t = self.torch.from_numpy(np.asarray(rows[0], np.float64).copy())
        out = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [o.numpy() for o in out]

    def barrier(self):
        for c in self.ctx:
            c.sync()
        self.dist.barrier()


def worker(rank, world, Np, nobs, q):
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import slam_amd as sg
    from conftest import sim_args
    from slam_amd import host
    from slam_amd.dist import DistFilter
    tp = host.make_tape(sim_args("example_webmap", "FASTSLAM2", Np, 3), max_obs=nobs)
    n = Np // world
    c = sg.SlamGpu(n, tp["nlm"], method=sg.FASTSLAM2, n_effective=int(0.75 * Np), rng_mode=sg.RNG_PHILOX, seed=9, math_mode=1,
                   first_particle=rank * n, n_particles_global=Np)
    f = DistFilter([c], GlooGather(c, rank, world))
    if os.environ.get("SLAM_DIST_PUSH"):
        assert f.use_push(fold=os.environ["SLAM_DIST_PUSH"] == "fold"), "a peer did not arrive at the flag handshake"
    for st in tp["steps"]:
        f.step(np.array(st["controls"], np.float32).reshape(-1, 3), tp["Q"], float(tp["dt"]), st["zf"], st["idf"], st["zn"], tp["R"])
    hist = f.history_fetch()
    d = f.download()[0]
    if os.environ.get("SLAM_DIST_PUSH"):
        assert f.collective_ok()
    f.close()
    q.put((rank, d, hist))
    dist.barrier()
    dist.destroy_process_group()


def main():
    import multiprocessing as mp
    Np, nobs, world = 4096, 60, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, world, Np, nobs, q)) for r in range(world)]
    for p in ps:
        p.start()
    got = dict()
    for _ in range(world):
        r, d, h = q.get(timeout=500)
        got[r] = (d, h)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0, p.exitcode
    import slam_amd as sg
    from conftest import sim_args
    from slam_amd import host
    tp = host.make_tape(sim_args("example_webmap", "FASTSLAM2", Np, 3), max_obs=nobs)
    s = sg.SlamGpu(Np, tp["nlm"], method=sg.FASTSLAM2, n_effective=int(0.75 * Np), rng_mode=sg.RNG_PHILOX, seed=9, math_mode=1)
    for st in tp["steps"]:
        s.step(np.array(st["controls"], np.float32).reshape(-1, 3), tp["Q"], float(tp["dt"]), st["zf"], st["idf"], st["zn"], tp["R"])
    href = s.history_fetch()
    ref = s.download()
    s.close()
    assert href[2].sum() >= 3
    for key in ("xv", "Pv", "w", "xf", "Pf"):
        cat = np.concatenate([got[r][0][key] for r in range(world)])
        assert np.array_equal(cat.view(np.uint32), ref[key].view(np.uint32)), key
    for r in range(world):
        assert np.allclose(got[r][1][0], href[0], rtol=0, atol=1e-12)
        assert np.array_equal(got[r][1][2], href[2])
    print("DIST_IPC_OK" + (" " + os.environ["SLAM_DIST_PUSH"] if os.environ.get("SLAM_DIST_PUSH") else ""))


if __name__ == "__main__":
    main()
