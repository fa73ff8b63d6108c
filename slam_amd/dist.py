"""Distributed FastSLAM: the particle set spans several GPUs and nothing migrates (include/slamgpu.h: slamgpu_dist_*).

Shard g holds the contiguous global particles [g*n, (g+1)*n).  Every context maps every other shard's state arrays
(peer access inside one process, hipIpc across processes).  Per observation step and shard:

  ONE launch    the queued predicts + the resampling stage of the PREVIOUS step (every shard scans the same all-gathered
                block totals => identical sum w, Neff, decision and ancestors, independent of the number of shards:
                core.cpp:718-824) + the per-particle update (fastslam2.cpp:107-168).  An ancestor on another GPU is read in
                place over xGMI; genealogy entries are global slot ids.
  ONE all-gather  of this step's block totals (8 B per 256 particles per shard), stream-ordered: RCCL through
                torch.distributed (`TorchGather`), or device copies between the contexts of one process (`LocalGather`).

The pose estimate (ParticleSLAMWrapper.cpp:56-77) is combined from the shards' raw partials when the history is fetched.
"""
import numpy as np

from . import capi


class LocalGather:
    """All shards are contexts of this process (logical shards on one GPU, or one process driving several GPUs)."""

    def __init__(self, contexts):
        self.ctx = contexts
        self.world = len(contexts)
        self.shards = list(range(self.world))

    def exchange_blobs(self, blobs):
        return list(blobs)

    def all_gather(self):
        bufs = [c.dist_totals() for c in self.ctx]
        for c in self.ctx:  # each context runs on its own stream: its totals must have landed before a peer copies them
            c.sync()
        for h, c in enumerate(self.ctx):
            for g in range(self.world):
                loc, _, n = bufs[g]
                c.dev_copy(bufs[h][1] + 4 * n * g, loc, 4 * n)

    def all_gather_rows(self, rows):
        return [np.asarray(r, np.float64) for r in rows]

    def barrier(self):
        for c in self.ctx:
            c.sync()


class _DevArray:
    """a raw device pointer as something torch.as_tensor can wrap without copying"""

    def __init__(self, ptr, nfloats):
        self.__cuda_array_interface__ = {"shape": (nfloats,), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


class TorchGather:
    """One shard per rank; torch.distributed with backend nccl (= RCCL over xGMI).  The context must have been created with
    external_stream = torch's current (non-default) stream: the collective is then ordered with the launches and the step
    loop never waits on the host."""

    def __init__(self, context, device):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.ctx = [context]
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.shards = [self.rank]
        self.device = device
        cur = torch.cuda.current_stream(device).cuda_stream
        if (context.stream() or 0) != cur or cur == 0:
            raise RuntimeError("TorchGather: create the context with external_stream=torch.cuda.current_stream().cuda_stream "
                               "on an explicit non-default stream (context %#x, torch %#x)" % (context.stream() or 0, cur))
        self._views = {}

    def exchange_blobs(self, blobs):
        out = [None] * self.world
        self.dist.all_gather_object(out, blobs[0])
        return out

    def all_gather(self):
        loc, gat, n = self.ctx[0].dist_totals()
        v = self._views.get((loc, gat))
        if v is None:  # two parities: two pairs of views, made once
            t = self.torch
            v = (t.as_tensor(_DevArray(gat, n * self.world), device=self.device), t.as_tensor(_DevArray(loc, n), device=self.device))
            self._views[(loc, gat)] = v
        self.dist.all_gather_into_tensor(v[0], v[1])

    def all_gather_rows(self, rows):
        t = self.torch.tensor(np.asarray(rows[0], np.float64), dtype=self.torch.float64, device=self.device)
        out = self.torch.zeros((self.world,) + tuple(t.shape), dtype=self.torch.float64, device=self.device)
        self.dist.all_gather_into_tensor(out, t)
        return list(out.cpu().numpy())

    def barrier(self):
        self.dist.barrier()


class NativeGather(TorchGather):
    """One shard per rank; the all-gather runs INSIDE slamgpu_dist_step (RCCL bound by libslamgpu itself:
    slamgpu_dist_comm_init), so a filter step is a single C call.  torch.distributed only carries the set-up (mapping
    blobs, the RCCL unique id) and the final combination of the estimate partials."""

    native = True

    def connect_comm(self):
        ids = [None]
        if self.rank == 0:
            try:
                ids = [capi.dist_comm_id()]
            except Exception as e:  # noqa: BLE001 (every rank learns about it below)
                ids = ["error: %s" % e]
        self.dist.broadcast_object_list(ids, src=0)
        if not isinstance(ids[0], bytes):
            raise RuntimeError("distributed set-up: no RCCL unique id (%s)" % ids[0])
        self.ctx[0].dist_comm_init(ids[0], self.world, self.rank)

    def all_gather(self):
        pass


class DistFilter:
    """FastSLAM{1,2}::predict / ::update / computeEstimatedPosition over a particle set distributed across contexts.

    `contexts`: the shards living in this process (all of them with LocalGather, one with TorchGather), created with
    first_particle = shard * n, n_particles_global = n_shards * n, rng_mode = RNG_PHILOX."""

    def __init__(self, contexts, gather, on_stage=None):
        """on_stage(name): called as each collective step of the set-up BEGINS (bench.py's breadcrumbs: a rank that blocks in
        hipIpcOpenMemHandle or ncclCommInitRank is then named with the step it blocks in)"""
        self.ctx, self.g = list(contexts), gather
        self.push = False
        stage = on_stage or (lambda name: None)
        self.G = gather.world
        self.n = getattr(self.ctx[0], "n_local", None) or self.ctx[0].N  # particles per shard
        # every step of the set-up is collective: a rank that fails must still take part in the exchange, and all ranks
        # leave together (an exception on one rank alone would leave the others waiting in a collective forever)
        mine, err = [], None
        stage("ipc_export")
        for c in self.ctx:
            try:
                mine.append(c.dist_export())
            except Exception as e:  # noqa: BLE001
                mine.append(None)
                err = err or e
        stage("ipc_exchange")
        blobs = gather.exchange_blobs(mine)
        if any(b is None for b in blobs):
            raise RuntimeError("distributed set-up: a shard could not export its state arrays (%s)" % (err or "on another rank"))
        stage("peer_mappings")
        for c, s in zip(self.ctx, gather.shards):
            try:
                c.dist_connect(self.G, s, blobs)
            except Exception as e:  # noqa: BLE001
                err = err or e
        if not self._agree(err is None):
            raise RuntimeError("distributed set-up: a shard could not map its peers (%s)" % (err or "on another rank"))
        if hasattr(gather, "connect_comm"):
            stage("communicator")
            gather.connect_comm()
        stage("setup_barrier")
        gather.barrier()

    def use_push(self, iters=20, fold=False):
        """switch to the push collective (slamgpu.h: SLAMGPU_DIST_PUSH; fold=True: SLAMGPU_DIST_FOLD, the barrier inside the
        next launch) after trying `iters` barriers; returns False (and stays with the gather) if a peer did not arrive.
        Collective; call between settled steps."""
        try:
            if len(self.ctx) == 1:
                _, ok = self.ctx[0].dist_handshake_test(iters)
            else:
                for c in self.ctx:
                    c.dist_handshake_enqueue(iters)
                ok = all(c.dist_collective_ok() for c in self.ctx)
        except capi.SlamGpuError:  # (no fine-grained flag words on this stack)
            ok = False
        if not self._agree(ok):
            return False
        for c in self.ctx:
            c.dist_set_collective(2 if fold else 1)
        self.push = True
        return True

    def use_gather(self):
        """back to the all-gather collective (between settled steps, every shard alike)"""
        for c in self.ctx:
            c.dist_set_collective(False)
        self.push = False

    def collective_ok(self):
        return self._agree(all(c.dist_collective_ok() for c in self.ctx))

    def _agree(self, ok):
        flags = self.g.exchange_blobs([bytes([1 if ok else 0])] * len(self.ctx))
        return all(f == bytes([1]) for f in flags)

    @classmethod
    def local(cls, n_shards, n_per_shard, max_landmarks, devices=None, **kw):
        kw.setdefault("rng_mode", capi.RNG_PHILOX)
        ctx = [capi.SlamGpu(n_per_shard, max_landmarks, first_particle=g * n_per_shard, n_particles_global=n_shards * n_per_shard,
                            device=(devices[g] if devices else 0), **kw) for g in range(n_shards)]
        return cls(ctx, LocalGather(ctx))

    def prepare_step(self, controls, Q, dt, zf, idf, zn, R, record_estimate=True):
        calls = [c.prepare_dist_step(controls, Q, dt, zf, idf, zn, R, record_estimate) for c in self.ctx]
        if getattr(self.g, "native", False) and len(calls) == 1:
            return calls[0]
        gather = (lambda: None) if self.push else self.g.all_gather

        def call():
            for f in calls:
                f()
            gather()
        return call

    def step(self, controls, Q, dt, zf, idf, zn, R, record_estimate=True):
        self.prepare_step(controls, Q, dt, zf, idf, zn, R, record_estimate)()

    def settle(self):
        """apply the pending resampling stage everywhere (collective); afterwards the contexts can be read"""
        for c in self.ctx:
            c.dist_settle()
        if not self.push:
            self.g.all_gather()
        self.g.barrier()

    def history_fetch(self):
        """(xyt[k,3], neff[k], resampled[k]) of the recorded steps, combined over the shards in shard order"""
        self.settle()
        raws = [c.shard_estimate_fetch_full() for c in self.ctx]
        k = min(len(r[0]) for r in raws)
        rows = self.g.all_gather_rows([r[0][:k] for r in raws])  # [G][k,4]
        xyt = np.zeros((k, 3))
        best = np.full(k, -np.inf)
        for r in rows:
            r = np.asarray(r).reshape(k, 4)
            xyt[:, 0] += r[:, 0]
            xyt[:, 1] += r[:, 1]
            better = r[:, 3] > best  # strict: ties keep the lowest global index (ParticleSLAMWrapper.cpp:60-68)
            xyt[better, 2] = r[better, 2]
            best = np.where(better, r[:, 3], best)
        xyt[:, :2] /= float(self.n * self.G)
        _, neff, res, st = raws[0]
        self.last_history_status = st[:k]
        return xyt, neff[:k], res[:k].astype(bool)  # (same shape as SlamGpu.history_fetch)

    def download(self, landmarks=True):
        """this process's shards, concatenated in shard order (settles first)"""
        self.settle()
        parts = [c.download(landmarks) for c in self.ctx]
        self.g.barrier()
        return parts

    def nf(self):
        return self.ctx[0].nf()

    def sync(self):
        for c in self.ctx:
            c.sync()

    def close(self):
        self.g.barrier()
        for c in self.ctx:
            c.close()
