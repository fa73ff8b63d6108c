"""Data-parallel FastSLAM over shards of the particle set (one shard per GPU / rank).

Shard g holds the contiguous global particles [g*n, (g+1)*n).  Predict and the per-particle observation
update are embarrassingly parallel; the only coupling is resampleParticles (core.cpp:718-824) and the pose
estimate (ParticleSLAMWrapper.cpp:56-77).  Per observation step:

  all-gather   per-256-particle totals of w and w^2          (8 B per 256 particles per shard)
  every shard  runs the same scan of the gathered totals  =>  identical sum w, Neff, decision and offspring
               boundaries K[0..G]; results do not depend on the number of shards
  all-to-all   (only when the resample fires) offspring records, (40 + 20*Nf) B each, to the shard that owns the
               output slot; with balanced weights most records stay on their shard
  all-gather   4 doubles per shard for the pose estimate

The collectives go through a small `Comm` interface: `TorchComm` = torch.distributed (backend "nccl" = RCCL over
xGMI on the GPU box, "gloo" in the CPU tests), `LocalComm` = several logical shards inside one process (tests,
single-GPU rehearsal).  An *engine* is one shard's compute: `GpuEngine` wraps a slamgpu context through the C ABI.
"""
import numpy as np

from . import capi


class GpuEngine:
    """One shard on one GPU through include/slamgpu.h."""

    is_gpu = True

    def __init__(self, shard, n_shards, n_per_shard, max_landmarks, **kw):
        self.shard, self.n_shards, self.n = shard, n_shards, n_per_shard
        self.ctx = capi.SlamGpu(n_per_shard, max_landmarks, first_particle=shard * n_per_shard,
                                n_particles_global=n_shards * n_per_shard, **kw)

    def predict(self, V, G, Q, dt, phi_true=0.0, noise2=None):
        self.ctx.predict(V, G, Q, dt, phi_true, noise2)

    def local_update(self, zf, idf, zn, R, normals, strata):
        self.ctx.shard_update(zf, idf, zn, R, normals, strata)

    def step_local(self, controls, Q, dt, zf, idf, zn, R, normals, strata):
        """the queued predicts + the per-particle update in one C-ABI call (slamgpu_shard_step)"""
        self.ctx.shard_step(controls, Q, dt, zf, idf, zn, R, normals, strata)

    def prepare_step_local(self, controls, Q, dt, zf, idf, zn, R, normals, strata):
        """step_local with the arguments marshalled once (SlamGpu.prepare_step): returns a callable"""
        return self.ctx.prepare_step(controls, Q, dt, zf, idf, zn, R, normals, strata, shard=True)

    def use_totals_buffer(self, comm, buf):
        """let the update kernel write the block totals straight into the collective's input buffer"""
        self.ctx.shard_set_totals_buffer(comm.ptr(buf))
        self._totals_direct = True

    def block_totals_into(self, comm, buf):
        if getattr(self, "_totals_direct", False):
            if not getattr(comm, "stream_ordered", False):
                self.ctx.sync()
            return
        t, nb = self.ctx.shard_block_totals()
        # a comm whose collectives are ordered on this context's stream needs no host wait here
        self.ctx.dev_copy(comm.ptr(buf), t, 8 * nb, not getattr(comm, "stream_ordered", False))

    def estimate_async(self):
        self.ctx.shard_estimate_async()

    def estimate_fetch(self):
        return self.ctx.shard_estimate_fetch()

    def nblocks(self):
        return self.n // 256

    def plan(self, comm, gtot, nb_global):
        return self.ctx.shard_plan(comm.ptr(gtot), nb_global, self.n_shards)

    def record_floats(self):
        return self.ctx.record_floats()

    def pack(self, comm, gtot, nb_global, plan, send):
        return self.ctx.shard_pack(comm.ptr(gtot), nb_global, self.n_shards, self.shard, plan, comm.ptr(send))

    def unpack(self, comm, recv, plan):
        self.ctx.shard_unpack(comm.ptr(recv), self.n_shards, self.shard, plan)

    def finish(self, plan):
        self.ctx.shard_finish(plan)

    def estimate_local(self):
        return self.ctx.shard_estimate()

    def sync(self):
        self.ctx.sync()

    def close(self):
        self.ctx.close()


class LocalComm:
    """All shards live in this process (logical shards).  Buffers come from the first engine's allocator: raw device
    pointers for GPU engines (same device), numpy arrays for CPU engines."""

    def __init__(self, engines):
        self.engines = engines
        self.gpu = getattr(engines[0], "is_gpu", False)
        self._keep = []

    def alloc(self, engine, nfloats):
        if self.gpu:
            p = engine.ctx.dev_alloc(4 * max(int(nfloats), 1))
            self._keep.append((engine, p))
            return p
        return np.zeros(max(int(nfloats), 1), np.float32)

    def free_all(self):
        for e, p in self._keep:
            e.ctx.dev_free(p)
        self._keep = []

    def ptr(self, buf):
        return buf if self.gpu else buf.ctypes.data

    def _copy(self, engine, dst, dst_off, src, src_off, nfloats):
        if nfloats <= 0:
            return
        if self.gpu:
            engine.ctx.dev_copy(dst + 4 * dst_off, src + 4 * src_off, 4 * nfloats)
        else:
            dst[dst_off:dst_off + nfloats] = src[src_off:src_off + nfloats]

    def all_gather(self, local_bufs, nfloats_each, global_bufs):
        """global_bufs[i][g*n:(g+1)*n] = local_bufs[g] for every local shard i."""
        for e in self.engines:  # each shard runs on its own stream: its data must have landed before a peer copies it
            e.sync()
        for i, e in enumerate(self.engines):
            for g in range(len(self.engines)):
                self._copy(e, global_bufs[i], g * nfloats_each, local_bufs[g], 0, nfloats_each)

    def all_to_all(self, send_bufs, send_counts, recv_bufs, recv_counts):
        """counts in floats; send_counts[g][d] floats go from shard g to shard d."""
        G = len(self.engines)
        for e in self.engines:
            e.sync()
        for d in range(G):
            roff = 0
            for g in range(G):
                soff = int(sum(send_counts[g][:d]))
                n = int(send_counts[g][d])
                assert n == int(recv_counts[d][g]), (g, d, n, recv_counts[d][g])
                self._copy(self.engines[d], recv_bufs[d], roff, send_bufs[g], soff, n)
                roff += n

    def all_gather_small(self, local_vecs):
        return [np.asarray(v, np.float64) for v in local_vecs]

    def barrier(self):
        pass


class TorchComm:
    """One shard per rank over torch.distributed (backend nccl == RCCL on ROCm; gloo on CPU)."""

    def __init__(self, device=None, stream_ordered=False):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device = device if device is not None else torch.device("cpu")
        # True when the engines launch on torch's current stream (slamgpu_config.external_stream): RCCL collectives
        # enqueued through torch are then ordered with the kernels without host synchronisation
        self.stream_ordered = stream_ordered
        self.engines = []  # registered by ShardedFilter

    def _is_gpu(self):
        return self.device.type == "cuda"

    def _before(self):
        # engines on streams of their own: what they enqueued (block totals, packed offspring) must have landed before
        # the collective, which torch orders on ITS current stream, may read it
        if self._is_gpu() and not self.stream_ordered:
            for e in self.engines:
                e.sync()

    def _after(self):
        # ... and the collective must have finished before the engines' kernels read what it produced
        if self._is_gpu() and not self.stream_ordered:
            self.torch.cuda.current_stream(self.device).synchronize()

    def check_engine(self, e):
        """stream_ordered promises that the engine launches on the stream torch orders the collectives on"""
        if self._is_gpu() and self.stream_ordered and getattr(e, "is_gpu", False):
            cur = self.torch.cuda.current_stream(self.device).cuda_stream
            if (e.ctx.stream() or 0) != cur or cur == 0:
                raise RuntimeError("TorchComm(stream_ordered=True): the engine's HIP stream (%#x) is not torch's current "
                                   "stream (%#x): create the context with external_stream=torch.cuda.current_stream().cuda_stream "
                                   "on an explicit non-default stream" % (e.ctx.stream() or 0, cur))

    def alloc(self, engine, nfloats):
        return self.torch.zeros(max(int(nfloats), 1), dtype=self.torch.float32, device=self.device)

    def free_all(self):
        pass

    def ptr(self, buf):
        return buf.data_ptr()

    def all_gather(self, local_bufs, nfloats_each, global_bufs):
        key = (local_bufs[0].data_ptr(), global_bufs[0].data_ptr(), nfloats_each)
        if getattr(self, "_ag_key", None) != key:  # the step loop gathers the same two buffers every time: keep the views
            self._ag_key = key
            self._ag_views = (global_bufs[0][: nfloats_each * self.world], local_bufs[0][:nfloats_each])
        self._before()
        self.dist.all_gather_into_tensor(*self._ag_views)
        self._after()

    def all_to_all(self, send_bufs, send_counts, recv_bufs, recv_counts):
        sc = [int(x) for x in send_counts[0]]
        rc = [int(x) for x in recv_counts[0]]
        self._before()
        self.dist.all_to_all_single(recv_bufs[0][: sum(rc)], send_bufs[0][: sum(sc)], output_split_sizes=rc, input_split_sizes=sc)
        self._after()

    def all_gather_small(self, local_vecs):
        t = self.torch.tensor(np.asarray(local_vecs[0], np.float64), dtype=self.torch.float64, device=self.device)
        out = self.torch.zeros(self.world * t.numel(), dtype=self.torch.float64, device=self.device)
        self.dist.all_gather_into_tensor(out, t)
        return list(out.cpu().numpy().reshape(self.world, -1))

    def barrier(self):
        self.dist.barrier()


class ShardedFilter:
    """FastSLAM{1,2}::predict / ::update / computeEstimatedPosition over shards.

    `engines` are the shards living in this process (one with TorchComm, all of them with LocalComm);
    `n_shards` is the global number of shards."""

    def __init__(self, engines, comm, n_shards):
        self.engines, self.comm, self.G = engines, comm, n_shards
        if hasattr(comm, "check_engine"):
            comm.engines = list(engines)
            for e in engines:
                comm.check_engine(e)
        self.nb_local = engines[0].nblocks()
        self.nb_global = self.nb_local * n_shards
        self.n = engines[0].n
        c = comm
        # block totals: each shard contributes one contiguous [w(nb) | w2(nb)] message; gathered shard-major
        self.loc = [c.alloc(e, 2 * self.nb_local) for e in engines]
        self.gtot = [c.alloc(e, 2 * self.nb_global) for e in engines]
        self.send = [None] * len(engines)
        self.recv = [None] * len(engines)
        self.cap = 0
        self.send_cap = [0] * len(engines)
        self.last_plan = None
        self.exchanged_records = 0
        for i, e in enumerate(engines):
            if hasattr(e, "use_totals_buffer"):
                e.use_totals_buffer(c, self.loc[i])

    def predict(self, V, G, Q, dt, phi_true=0.0, noise2=None):
        for i, e in enumerate(self.engines):
            e.predict(V, G, Q, dt, phi_true, None if noise2 is None else noise2[i])

    def _ensure(self, fields, plans):
        # a shard receives exactly n records but may SEND more than n (a heavy shard spawns more offspring)
        need_recv = self.n * fields
        if need_recv > self.cap:
            self.recv = [self.comm.alloc(e, need_recv) for e in self.engines]
            self.cap = need_recv
        for i, e in enumerate(self.engines):
            need = int(plans[i].K[e.shard + 1] - plans[i].K[e.shard]) * fields
            if need > self.send_cap[i]:
                self.send[i] = self.comm.alloc(e, int(need * 1.25) + fields)
                self.send_cap[i] = int(need * 1.25) + fields

    def step(self, controls, Q, dt, zf, idf, zn, R, normals=None, strata=None, record_estimate=True):
        """One whole filter step: the predicts of `controls` ([k,3]: V, G, phi_true), the update with its global
        resampling stage and (optionally) the asynchronous pose estimate."""
        for i, e in enumerate(self.engines):
            nm = None if normals is None else normals[i]
            if hasattr(e, "step_local"):
                e.step_local(controls, Q, dt, zf, idf, zn, R, nm, strata)
            else:
                for (V, G, phi) in np.asarray(controls, np.float32).reshape(-1, 3):
                    e.predict(float(V), float(G), Q, dt, float(phi))
                e.local_update(zf, idf, zn, R, nm, strata)
        plan = self._resample_stage()
        if record_estimate:
            self.estimate_async()
        return plan

    def prepare_step(self, controls, Q, dt, zf, idf, zn, R, normals=None, strata=None, record_estimate=True):
        """step() with the per-shard arguments marshalled once, outside a driver's timed loop; returns a callable that
        performs the step and returns its plan."""
        locs = []
        for i, e in enumerate(self.engines):
            nm = None if normals is None else normals[i]
            if hasattr(e, "prepare_step_local"):
                locs.append(e.prepare_step_local(controls, Q, dt, zf, idf, zn, R, nm, strata))
            else:
                locs.append(lambda e=e, nm=nm: (self._predicts(e, controls, Q, dt), e.local_update(zf, idf, zn, R, nm, strata)))

        def call():
            for f in locs:
                f()
            plan = self._resample_stage()
            if record_estimate:
                self.estimate_async()
            return plan
        return call

    @staticmethod
    def _predicts(e, controls, Q, dt):
        for (V, G, phi) in np.asarray(controls, np.float32).reshape(-1, 3):
            e.predict(float(V), float(G), Q, dt, float(phi))

    def update(self, zf, idf, zn, R, normals=None, strata=None):
        """normals: per local shard [n,3] arrays (tape mode) or None; strata: global [N] (tape mode) or None."""
        for i, e in enumerate(self.engines):
            e.local_update(zf, idf, zn, R, None if normals is None else normals[i], strata)
        return self._resample_stage()

    def _resample_stage(self):
        E, c = self.engines, self.comm
        for i, e in enumerate(E):
            e.block_totals_into(c, self.loc[i])
        c.all_gather(self.loc, 2 * self.nb_local, self.gtot)
        plans = [e.plan(c, self.gtot[i], self.nb_global) for i, e in enumerate(E)]
        plan = plans[0]
        if plan.resampled:
            fields = E[0].record_floats()
            self._ensure(fields, plans)
            sc, rc = [], []
            for i, e in enumerate(E):
                s_, r_ = e.pack(c, self.gtot[i], self.nb_global, plans[i], self.send[i])
                sc.append(s_ * fields)
                rc.append(r_ * fields)
                self.exchanged_records += int(s_.sum())
            # offspring whose output slot is on their own shard never leave it; skip the collective when the plan
            # (identical on every shard) says that nothing crosses a shard boundary
            if any(int(plan.K[r]) != r * self.n for r in range(1, self.G)):
                c.all_to_all(self.send, sc, self.recv, rc)
            for i, e in enumerate(E):
                e.unpack(c, self.recv[i], plans[i])
        for i, e in enumerate(E):
            e.finish(plans[i])
        self.last_plan = plan
        return plan

    def estimate(self):
        """mean x, mean y over all particles; heading of the first particle with the strictly greatest weight."""
        parts = self.comm.all_gather_small([e.estimate_local() for e in self.engines])
        if len(parts) != self.G:  # LocalComm returns one vector per local shard == all shards
            raise RuntimeError("estimate: expected %d shard partials, got %d" % (self.G, len(parts)))
        sx = sum(p[0] for p in parts)
        sy = sum(p[1] for p in parts)
        best_w, best_t = -np.inf, 0.0
        for p in parts:  # shard order == particle order: first strict maximum wins
            if p[3] > best_w:
                best_w, best_t = p[3], p[2]
        N = self.n * self.G
        return np.array([sx / N, sy / N, best_t])

    def estimate_async(self):
        """Queue this step's local estimate partials on every shard (no synchronisation, no collective)."""
        for e in self.engines:
            e.estimate_async()

    def estimate_fetch(self):
        """All queued estimates: one synchronisation and one all-gather for the whole history -> [steps, 3]."""
        local = [e.estimate_fetch() for e in self.engines]  # each [T, 4]: sum x, sum y, heading, max w
        T = local[0].shape[0]
        parts = self.comm.all_gather_small([l.ravel() for l in local])
        parts = [np.asarray(p).reshape(T, 4) for p in parts]
        N = self.n * self.G
        out = np.zeros((T, 3))
        for t in range(T):
            best_w = -np.inf
            for p in parts:
                out[t, 0] += p[t, 0]
                out[t, 1] += p[t, 1]
                if p[t, 3] > best_w:
                    best_w, out[t, 2] = p[t, 3], p[t, 2]
        out[:, :2] /= N
        return out

    def sync(self):
        for e in self.engines:
            e.sync()

    def close(self):
        self.comm.free_all()
        for e in self.engines:
            e.close()
