"""The N>1 path on CPU: the distributed orchestration (slam_amd/dist.py: DistFilter) over logical shards and over
torch.distributed (gloo, world_size 2), with the oracle standing in for a distributed context (tests/cpu_dist_engine.py).
Results must not depend on the number of shards, must equal what the exchange-based orchestration (slam_amd/sharded.py)
computes, and the estimate history must combine to the single-shard one."""
import os
import sys

import numpy as np

from conftest import sim_args

f32 = np.float32
N = 1024
NOBS = 14


def make_tape(nobs):
    from slam_amd import host
    return host.make_tape(sim_args("example_webmap", "FASTSLAM2", N, 7), max_obs=nobs)


def make_algo():
    from oracle import orc
    return orc.Algo(2, 0, 0, 1, int(0.75 * N), 4.0, 0.017453292519943)


def drive(flt, tape):
    for st in tape["steps"]:
        flt.step(np.array(st["controls"], f32).reshape(-1, 3), tape["Q"], float(tape["dt"]), st["zf"], st["idf"], st["zn"], tape["R"])
    hist = flt.history_fetch()
    parts = flt.download()
    return hist, parts


def run_local(oracle, G, tape):
    from slam_amd.dist import DistFilter
    from cpu_dist_engine import CpuDistContext, CpuLocalGather
    ctx = [CpuDistContext(oracle, g, G, N // G, tape["nlm"], make_algo()) for g in range(G)]
    flt = DistFilter(ctx, CpuLocalGather(ctx))
    hist, parts = drive(flt, tape)
    moved = sum(getattr(c, "moved", 0) for c in ctx)
    cat = {k: np.concatenate([p[k] for p in parts]) for k in ("xv", "Pv", "w", "xf", "Pf")}
    flt.close()
    return hist, cat, moved


def test_shard_count_invariance(oracle):
    tape = make_tape(NOBS)
    href, ref, _ = run_local(oracle, 1, tape)
    assert href[2].any() and not href[2].all() and len(href[0]) == NOBS
    for G in (2, 4):
        h, got, moved = run_local(oracle, G, tape)
        assert moved > 0  # ancestors were read across shard boundaries
        assert np.array_equal(h[1], href[1]) and np.array_equal(h[2], href[2])
        assert np.allclose(h[0], href[0], rtol=0, atol=1e-12)
        for k in ref:
            assert np.array_equal(ref[k], got[k]), (G, k)


def test_matches_the_exchange_orchestration(oracle):
    """slam_amd/sharded.py (plan / pack / all-to-all / unpack) and slam_amd/dist.py (nothing migrates) are two
    orchestrations of the same filter.  Step by step on the same tape: identical until the first resample has travelled
    through the exchange engine's packed records (they carry the lower triangle of Pv, i.e. symmetrise it: an ulp in a
    weight now and then, hence an ancestor at a cumulative-sum boundary, hence shifted slots); same decisions, Neff and
    mean pose throughout."""
    from slam_amd.dist import DistFilter
    from slam_amd.sharded import LocalComm, ShardedFilter
    from cpu_dist_engine import CpuDistContext, CpuLocalGather
    from cpu_shard_engine import CpuEngine
    tape = make_tape(NOBS)
    eng = [CpuEngine(oracle, g, 2, N // 2, tape["nlm"], make_algo()) for g in range(2)]
    flt = ShardedFilter(eng, LocalComm(eng), 2)
    ctx = [CpuDistContext(oracle, g, 2, N // 2, tape["nlm"], make_algo()) for g in range(2)]
    df = DistFilter(ctx, CpuLocalGather(ctx))
    resamples, neffs = 0, []
    for st in tape["steps"]:
        for (V, Gs, phi) in st["controls"]:
            flt.predict(V, Gs, tape["Q"], float(tape["dt"]), phi)
        plan = flt.update(st["zf"], st["idf"], st["zn"], tape["R"])
        df.step(np.array(st["controls"], f32).reshape(-1, 3), tape["Q"], float(tape["dt"]), st["zf"], st["idf"], st["zn"], tape["R"])
        df.settle()
        a = np.concatenate([e.state()["xv"] for e in eng])
        b = np.concatenate([c.P.get()["xv"] for c in ctx])
        if resamples + int(plan.resampled) <= 1:
            assert np.abs(a - b).max() <= 1e-6
        assert np.abs(a.mean(axis=0) - b.mean(axis=0)).max() < 2e-2
        resamples += int(plan.resampled)
        neffs.append(float(plan.neff))
    h = df.history_fetch()
    assert resamples >= 2 and int(h[2].sum()) == resamples
    assert np.allclose(h[1][:7], neffs[:7], rtol=1e-5) and np.allclose(h[1], neffs, rtol=0.25)
    flt.close()
    df.close()


def _gloo_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import torch.distributed as dist
    from oracle import orc
    from slam_amd.dist import DistFilter
    from cpu_dist_engine import CpuDistContext, CpuGlooGather
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tape = make_tape(NOBS)
    c = CpuDistContext(orc.Oracle(), rank, world, N // world, tape["nlm"], make_algo())
    flt = DistFilter([c], CpuGlooGather(c, rank, world))
    hist, parts = drive(flt, tape)
    flt.close()
    q.put((rank, hist, parts[0]))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world2_matches_single_shard(oracle):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        r, h, d = q.get(timeout=240)
        res[r] = (h, d)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    href, ref, _ = run_local(oracle, 1, make_tape(NOBS))
    for r in range(2):
        h = res[r][0]
        assert np.array_equal(h[1], href[1]) and np.array_equal(h[2], href[2])
        assert np.allclose(h[0], href[0], rtol=0, atol=1e-12)
    for k in ("xv", "Pv", "w", "xf", "Pf"):
        assert np.array_equal(np.concatenate([res[0][1][k], res[1][1][k]]), ref[k]), k
