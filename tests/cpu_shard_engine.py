"""TEST-ONLY shard engine on the CPU oracle, with the same interface as slam_amd.sharded.GpuEngine, so that the
sharded orchestration (slam_amd/sharded.py: all-gather of block totals, common plan, all-to-all of offspring,
estimate combine) can be exercised over gloo / logical shards without a GPU.

The per-particle arithmetic is the oracle's; the resampling bookkeeping restates the device definition:
float32 in-block (256) inclusive prefixes and block totals, a float64 scan of the block totals, ancestor
= first particle whose cumulative weight exceeds stratum * W, strata = caller's tape or Philox (gid + u)/N.
"""
import numpy as np

from oracle import orc

f32 = np.float32
BLOCK = 256


class Plan:
    def __init__(self):
        self.wsum = self.wsq = 0.0
        self.neff = f32(0)
        self.resampled = 0
        self.K = None


class CpuEngine:
    is_gpu = False

    def __init__(self, oracle, shard, n_shards, n_per_shard, max_landmarks, algo, rng_mode=1, seed=7):
        assert n_per_shard % BLOCK == 0
        self.O, self.shard, self.n_shards, self.n = oracle, shard, n_shards, n_per_shard
        self.N = n_shards * n_per_shard
        self.first = shard * n_per_shard
        self.algo = algo
        self.P = oracle.particles(n_per_shard, max_landmarks)
        st = self.P.get()
        st["w"][:] = f32(1.0 / f32(self.N))
        self.P.set(st)
        self.rng_mode, self.seed, self.step, self.ctl = rng_mode, seed, 0, 0
        self.strata = None
        self.lcum = None

    # ---- engine interface ----
    def predict(self, V, G, Q, dt, phi_true=0.0, noise2=None):
        self.ctl += 1
        if self.algo.add_predict_noise and noise2 is None:
            noise2 = self.O.philox_predict_tape(self.seed, self.ctl, self.first, self.n)
        self.P.predict(self.algo, V, G, Q, dt, phi_true, None if noise2 is None else np.ascontiguousarray(noise2, f32))

    def local_update(self, zf, idf, zn, R, normals, strata):
        self.step += 1
        if normals is None:
            normals, _ = self.O.philox_update_tape(self.seed, self.step, self.first, self.n, self.N)
        self.strata = None if strata is None else np.asarray(strata, f32)
        self.P.update_local(self.algo, zf, idf, zn, R, np.ascontiguousarray(normals, f32))
        w = self.P.get()["w"].reshape(-1, BLOCK)
        lc = np.zeros_like(w)
        acc = np.zeros(w.shape[0], f32)
        for j in range(BLOCK):  # sequential float32 prefix inside each block
            acc = (acc + w[:, j]).astype(f32)
            lc[:, j] = acc
        self.lcum = lc
        self.blk_w = lc[:, -1].copy()
        self.blk_w2 = np.array([np.sum((w[b] * w[b]).astype(f32), dtype=f32) for b in range(w.shape[0])], f32)

    def nblocks(self):
        return self.n // BLOCK

    def block_totals_into(self, comm, buf):
        nb = self.nblocks()
        self._np(comm, buf)[:nb] = self.blk_w
        self._np(comm, buf)[nb:2 * nb] = self.blk_w2

    @staticmethod
    def _np(comm, buf):
        return buf if isinstance(buf, np.ndarray) else buf.numpy()

    def _sel(self, k0, cnt):
        if self.strata is not None:
            return self.strata[k0:k0 + cnt].astype(np.float64)
        _, sel = self.O.philox_update_tape(self.seed, self.step, k0, cnt, self.N, want_normals=False)
        return sel.astype(np.float64)

    def plan(self, comm, gtot, nb_global):
        nbl0 = nb_global // self.n_shards
        g = self._np(comm, gtot)[:2 * nb_global].reshape(self.n_shards, 2, nbl0)  # shard-major [w | w2]
        gw = g[:, 0, :].ravel().astype(np.float64)
        gw2 = g[:, 1, :].ravel().astype(np.float64)
        off = np.concatenate([[0.0], np.cumsum(gw)])
        p = Plan()
        p.wsum, p.wsq = float(off[-1]), float(np.sum(gw2))
        p.neff = f32((p.wsum * p.wsum) / p.wsq)
        p.resampled = int(bool(self.algo.resample) and p.neff < f32(self.algo.n_effective))
        target = self._sel(0, self.N) * p.wsum
        nbl = nb_global // self.n_shards
        p.K = [0] + [int(np.searchsorted(target, off[r * nbl], side="left")) for r in range(1, self.n_shards)] + [self.N]
        self.off = off
        return p

    def record_floats(self):
        return 10 + 5 * self.P.nf()

    def _records(self):
        st = self.P.get()
        nf = st["nf"]
        P = st["Pv"]
        rec = [st["xv"][:, 0], st["xv"][:, 1], st["xv"][:, 2], P[:, 0, 0], P[:, 1, 0], P[:, 1, 1], P[:, 2, 0], P[:, 2, 1], P[:, 2, 2],
               np.zeros(self.n, f32)]
        for l in range(nf):
            rec += [st["xf"][:, l, 0], st["xf"][:, l, 1], st["Pf"][:, l, 0, 0], st["Pf"][:, l, 1, 0], st["Pf"][:, l, 1, 1]]
        return np.stack(rec).astype(f32)  # [fields][n]

    def pack(self, comm, gtot, nb_global, plan, send):
        """Offspring whose output slot is on this shard are kept (self._local); the rest goes to the send buffer in
        per-destination blocks — the same split the device pack kernel makes."""
        k_lo, k_hi = plan.K[self.shard], plan.K[self.shard + 1]
        G, n = self.n_shards, self.n
        sc = np.array([max(0, min(k_hi, (d + 1) * n) - max(k_lo, d * n)) if d != self.shard else 0 for d in range(G)], np.int64)
        rc = np.array([max(0, min(plan.K[d + 1], (self.shard + 1) * n) - max(plan.K[d], self.shard * n)) if d != self.shard else 0
                       for d in range(G)], np.int64)
        self._local = None
        if k_hi > k_lo:
            target = self._sel(k_lo, k_hi - k_lo) * plan.wsum
            fb = self.first // BLOCK
            cum = (self.off[fb:fb + self.nblocks(), None] + self.lcum.astype(np.float64)).ravel()
            anc = np.minimum(np.searchsorted(cum, target, side="right"), self.n - 1)
            rec = self._records()[:, anc]  # [fields][cnt]
            rec[9] = (anc + self.first).astype(np.int32).view(f32)
            out = self._np(comm, send) if send is not None else None
            fields = rec.shape[0]
            pos = 0
            for d in range(G):
                a, b = max(k_lo, d * n) - k_lo, min(k_hi, (d + 1) * n) - k_lo
                if b <= a:
                    continue
                if d == self.shard:
                    self._local = (max(k_lo, d * n) - self.first, rec[:, a:b].copy())
                else:
                    out[pos * fields:(pos + (b - a)) * fields] = rec[:, a:b].ravel()
                    pos += b - a
        return sc, rc

    def unpack(self, comm, recv, plan):
        nf = self.P.nf()
        fields = 10 + 5 * nf
        n = self.n
        lo = [min(max(plan.K[s] - self.first, 0), n) for s in range(self.n_shards + 1)]
        rec = np.zeros((fields, n), f32)
        pos = 0
        for s in range(self.n_shards):
            cnt = lo[s + 1] - lo[s]
            if cnt <= 0:
                continue
            if s == self.shard:
                o, r = self._local
                assert o == lo[s] and r.shape[1] == cnt
                rec[:, lo[s]:lo[s + 1]] = r
            else:
                buf = self._np(comm, recv)
                rec[:, lo[s]:lo[s + 1]] = buf[pos * fields:(pos + cnt) * fields].reshape(fields, cnt)
                pos += cnt
        st = dict(nf=nf, xv=np.stack([rec[0], rec[1], rec[2]], 1), w=np.full(n, f32(1.0) / f32(self.N), f32))
        Pv = np.zeros((n, 3, 3), f32)
        Pv[:, 0, 0], Pv[:, 1, 0], Pv[:, 1, 1], Pv[:, 2, 0], Pv[:, 2, 1], Pv[:, 2, 2] = rec[3], rec[4], rec[5], rec[6], rec[7], rec[8]
        Pv[:, 0, 1], Pv[:, 0, 2], Pv[:, 1, 2] = rec[4], rec[6], rec[7]
        st["Pv"] = Pv
        xf = np.zeros((n, nf, 2), f32)
        Pf = np.zeros((n, nf, 2, 2), f32)
        for l in range(nf):
            r = rec[10 + 5 * l:15 + 5 * l]
            xf[:, l, 0], xf[:, l, 1] = r[0], r[1]
            Pf[:, l, 0, 0], Pf[:, l, 1, 0], Pf[:, l, 0, 1], Pf[:, l, 1, 1] = r[2], r[3], r[3], r[4]
        st["xf"], st["Pf"] = xf, Pf
        self.keep = rec[9].view(np.int32).copy()
        self.P.set(st)

    def finish(self, plan):
        if not plan.resampled:
            st = self.P.get()
            st["w"] = (st["w"] / f32(plan.wsum)).astype(f32)
            self.P.set(st)

    def estimate_local(self):
        st = self.P.get()
        i = int(np.argmax(st["w"]))  # first maximum
        return np.array([st["xv"][:, 0].astype(np.float64).sum(), st["xv"][:, 1].astype(np.float64).sum(), st["xv"][i, 2], st["w"][i]])

    def state(self):
        return self.P.get()

    def sync(self):
        pass

    def close(self):
        self.P.close()
