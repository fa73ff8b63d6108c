"""TEST-ONLY stand-in for a distributed slamgpu context (slam_amd.dist.DistFilter's context interface) on the CPU oracle,
so that the distributed orchestration -- blob exchange, one "launch" + one all-gather per step, the resampling stage of
step t applied at the head of step t+1 from the gathered totals, ancestors read out of other shards' state, estimate
partials combined at fetch time -- runs over logical shards and over torch.distributed/gloo without a GPU.

"Peer-mapped memory" is emulated by the gather object: after each step it hands every context a snapshot of every
shard's state (the GPU contexts read the same data in place over xGMI).  Per-particle arithmetic and the block-structured
resampling definition are cpu_shard_engine.CpuEngine's (float32 in-block prefixes, float64 scan of the block totals).
"""
import numpy as np

from cpu_shard_engine import BLOCK, CpuEngine, f32


class CpuDistContext(CpuEngine):
    def __init__(self, oracle, shard, n_shards, n_per_shard, max_landmarks, algo, seed=7):
        super().__init__(oracle, shard, n_shards, n_per_shard, max_landmarks, algo, rng_mode=1, seed=seed)
        self.n_local = n_per_shard
        self.pending = False       # a resampling stage (of the last update) not applied yet
        self.gtot = None           # gathered [G][2][nb] block totals of the last update
        self.peers = None          # gathered snapshots: [G] dicts (state + lcum)
        self.hist = []             # (raw4, neff, resampled) per recorded step
        self.record_next = False
        self.connected = False

    # ---- what DistFilter calls ----
    def dist_export(self):
        return bytes([self.shard])

    def dist_connect(self, n_shards, shard, blobs):
        assert n_shards == self.n_shards and shard == self.shard and [b[0] for b in blobs] == list(range(n_shards))
        self.connected = True

    def snapshot(self):
        st = self.P.get()
        return dict(st, lcum=self.lcum)

    def local_totals(self):
        return np.concatenate([self.blk_w, self.blk_w2]).astype(f32)

    def set_gathered(self, totals, snaps):
        self.gtot, self.peers = totals, snaps

    def _apply_pending(self):
        """the resampling stage of the last update (core.cpp:718-824) from the gathered totals, then its estimate partial"""
        if not self.pending:
            return
        G, nb = self.n_shards, self.nblocks()
        g = np.asarray(self.gtot, f32).reshape(G, 2, nb)
        gw = g[:, 0, :].ravel().astype(np.float64)
        off = np.concatenate([[0.0], np.cumsum(gw)])
        W, Q = float(off[-1]), float(np.sum(g[:, 1, :].ravel().astype(np.float64)))
        neff = f32((W * W) / Q)
        res = bool(self.algo.resample) and neff < f32(self.algo.n_effective)
        st = self.P.get()
        if res:
            target = self._sel(self.first, self.n) * W
            lc = np.concatenate([p["lcum"].astype(np.float64).ravel() for p in self.peers])
            cum = (np.repeat(off[:-1], BLOCK) + lc)
            anc = np.minimum(np.searchsorted(cum, target, side="right"), self.N - 1)
            for key in ("xv", "Pv", "xf", "Pf"):
                allv = np.concatenate([p[key] for p in self.peers])
                st[key] = allv[anc].copy()
            st["w"] = np.full(self.n, f32(1.0) / f32(self.N), f32)
            self.moved = int(np.count_nonzero(anc // self.n != self.shard))
        else:
            st["w"] = (st["w"] / f32(W)).astype(f32)
        self.P.set(st)
        if self.record_next:
            i = int(np.argmax(st["w"]))
            raw = np.array([st["xv"][:, 0].astype(np.float64).sum(), st["xv"][:, 1].astype(np.float64).sum(), st["xv"][i, 2], st["w"][i]])
            self.hist.append((raw, neff, res))
        self.pending = False

    def prepare_dist_step(self, controls, Q, dt, zf, idf, zn, R, record_estimate=True):
        def call():
            assert self.connected
            self._apply_pending()
            for (V, G, phi) in np.asarray(controls, f32).reshape(-1, 3):
                self.predict(float(V), float(G), Q, dt, float(phi))
            self.local_update(zf, idf, zn, R, None, None)
            self.pending, self.record_next = True, bool(record_estimate)
        return call

    def dist_settle(self):
        self._apply_pending()

    def shard_estimate_fetch_full(self):
        assert not self.pending
        h, self.hist = self.hist, []
        k = len(h)
        return (np.array([x[0] for x in h]).reshape(k, 4), np.array([x[1] for x in h], f32), np.array([x[2] for x in h], np.int32),
                np.zeros(k, np.int32))

    def download(self, landmarks=True):
        assert not self.pending
        return self.P.get()

    def nf(self):
        return self.P.nf()


class CpuLocalGather:
    """all shards in this process"""

    def __init__(self, contexts):
        self.ctx, self.world, self.shards = contexts, len(contexts), list(range(len(contexts)))

    def exchange_blobs(self, blobs):
        return list(blobs)

    def all_gather(self):
        tot = np.stack([c.local_totals() for c in self.ctx])
        snaps = [c.snapshot() for c in self.ctx]
        for c in self.ctx:
            c.set_gathered(tot, snaps)

    def all_gather_rows(self, rows):
        return [np.asarray(r, np.float64) for r in rows]

    def barrier(self):
        pass


class CpuGlooGather:
    """one shard per rank over torch.distributed (gloo)"""

    def __init__(self, context, rank, world):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.ctx, self.rank, self.world, self.shards = [context], rank, world, [rank]

    def exchange_blobs(self, blobs):
        out = [None] * self.world
        self.dist.all_gather_object(out, blobs[0])
        return out

    def all_gather(self):
        c = self.ctx[0]
        t = self.torch.from_numpy(c.local_totals().copy())
        out = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)                       # the collective of the real path: the block totals
        snaps = [None] * self.world
        self.dist.all_gather_object(snaps, c.snapshot())   # emulation of the peer mapping (test plumbing only)
        c.set_gathered(np.stack([o.numpy() for o in out]), snaps)

    def all_gather_rows(self, rows):
        t = self.torch.from_numpy(np.asarray(rows[0], np.float64).copy())
        out = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [o.numpy() for o in out]

    def barrier(self):
        self.dist.barrier()
