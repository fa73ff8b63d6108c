"""TEST INFRASTRUCTURE — ctypes bindings for the CPU oracle (oracle/liboracle.so) and, where it has been
built (authoring container only), the reference's own objects (oracle/_ref/libslamref.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libslamref.so")
# the same objects built with -DJACOBIAN_ACCELERATOR and tests/cabi/accel_shim.h: computeJacobians runs on the GPU
REF_ACCEL_SO = os.path.join(HERE, "_ref", "libslamref_accel.so")

f32 = np.float32
fp = C.POINTER(C.c_float)
ip = C.POINTER(C.c_int)


def _p(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])


def build_ref():
    subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


def have_ref():
    return os.path.exists(REF_SO)


def _argv(args):
    arr = (C.c_char_p * (len(args) + 1))(b"slam-backend", *[str(a).encode() for a in args])
    return len(args) + 1, arr


class _FuncLib:
    """Function-level API shared by the oracle (prefix orc_) and the reference driver (prefix ref_)."""

    def __init__(self, path, prefix):
        self.lib = C.CDLL(path)
        self.prefix = prefix
        L = self.lib
        g = lambda n: getattr(L, prefix + n)
        g("trig_offset").restype = C.c_float
        g("trig_offset").argtypes = [C.c_float]
        g("gauss_evaluate").restype = C.c_float
        g("gauss_evaluate").argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        g("compute_jacobians").argtypes = [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 4
        g("cholesky_update2").argtypes = [C.c_void_p] * 5
        g("observe_heading").argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float]
        g("add_feature").argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]

    def trig_offset(self, a):
        return f32(getattr(self.lib, self.prefix + "trig_offset")(C.c_float(a)))

    def compute_jacobians(self, xv, R, xf, Pf):
        n = xf.shape[0]
        zp = np.zeros((n, 2), f32)
        Hv = np.zeros((n, 2, 3), f32)
        Hf = np.zeros((n, 2, 2), f32)
        Sf = np.zeros((n, 2, 2), f32)
        getattr(self.lib, self.prefix + "compute_jacobians")(_p(xv), _p(R), _p(xf), _p(Pf), n, _p(zp), _p(Hv), _p(Hf), _p(Sf))
        return zp, Hv, Hf, Sf

    def gauss_evaluate(self, v, S, logflag=0):
        D = v.shape[0]
        return f32(getattr(self.lib, self.prefix + "gauss_evaluate")(_p(v), _p(S), D, logflag))

    def cholesky_update2(self, x, P, v, R, H):
        x = x.copy()
        P = P.copy()
        getattr(self.lib, self.prefix + "cholesky_update2")(_p(x), _p(P), _p(v), _p(R), _p(H))
        return x, P

    def observe_heading(self, xv, Pv, phi, sigma):
        xv = xv.copy()
        Pv = Pv.copy()
        getattr(self.lib, self.prefix + "observe_heading")(_p(xv), _p(Pv), C.c_float(phi), C.c_float(sigma))
        return xv, Pv

    def add_feature(self, xv, zn, R):
        n = zn.shape[0]
        xf = np.zeros((n, 2), f32)
        Pf = np.zeros((n, 2, 2), f32)
        getattr(self.lib, self.prefix + "add_feature")(_p(xv), _p(zn), n, _p(R), _p(xf), _p(Pf))
        return xf, Pf


class Sim:
    """Whole-simulation handle; same surface for oracle and reference."""

    def __init__(self, lib, prefix, args):
        self.L = lib
        self.pfx = prefix
        g = lambda n: getattr(lib, prefix + n)
        g("sim_create").restype = C.c_void_p
        argc, argv = _argv(args)
        self._argv = argv
        self.h = C.c_void_p(g("sim_create")(argc, argv))
        if not self.h:
            raise RuntimeError("sim_create failed")
        g("sim_step").argtypes = [C.c_void_p]
        g("sim_control").argtypes = [C.c_void_p]
        g("sim_observe").argtypes = [C.c_void_p]
        g("sim_observe").restype = None
        g("sim_true").argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        g("sim_last_obs").argtypes = [C.c_void_p] * 8
        g("sim_destroy").argtypes = [C.c_void_p]
        g("sim_nlandmarks").argtypes = [C.c_void_p]
        self.nlm = g("sim_nlandmarks")(self.h)

    def step(self):
        return getattr(self.L, self.pfx + "sim_step")(self.h)

    def control(self):
        return getattr(self.L, self.pfx + "sim_control")(self.h)

    def observe(self):
        getattr(self.L, self.pfx + "sim_observe")(self.h)

    def true_pose(self):
        x = np.zeros(3, f32)
        vg = np.zeros(2, f32)
        getattr(self.L, self.pfx + "sim_true")(self.h, _p(x), _p(vg))
        return x, vg

    def last_obs(self):
        zf = np.zeros((self.nlm, 2), f32)
        zn = np.zeros((self.nlm, 2), f32)
        z = np.zeros((self.nlm, 2), f32)
        idf = np.zeros(self.nlm, np.int32)
        vis = np.zeros(self.nlm, np.int32)
        n = C.c_int()
        nz = C.c_int()
        m = getattr(self.L, self.pfx + "sim_last_obs")(self.h, _p(zf), _p(idf), _p(zn), C.byref(n), _p(z), _p(vis), C.byref(nz))
        return dict(zf=zf[:m].copy(), idf=idf[:m].copy(), zn=zn[: n.value].copy(), z=z[: nz.value].copy(), vis=vis[: nz.value].copy())

    def close(self):
        if self.h:
            getattr(self.L, self.pfx + "sim_destroy")(self.h)
            self.h = None


class RefSim(Sim):
    def __init__(self, ref, args):
        super().__init__(ref.lib, "ref_", args)
        L = self.L
        L.ref_sim_nparticles.argtypes = [C.c_void_p]
        L.ref_sim_nf.argtypes = [C.c_void_p]
        L.ref_sim_estimate.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_sim_get_particles.argtypes = [C.c_void_p] * 6
        L.ref_sim_ekf_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        self.N = L.ref_sim_nparticles(self.h)

    def nf(self):
        return self.L.ref_sim_nf(self.h)

    def estimate(self):
        e = np.zeros(3, np.float64)
        self.L.ref_sim_estimate(self.h, _p(e))
        return e

    def particles(self):
        nf, N = self.nf(), self.N
        xv = np.zeros((N, 3), f32)
        Pv = np.zeros((N, 3, 3), f32)
        w = np.zeros(N, f32)
        xf = np.zeros((N, nf, 2), f32)
        Pf = np.zeros((N, nf, 2, 2), f32)
        self.L.ref_sim_get_particles(self.h, _p(xv), _p(Pv), _p(w), _p(xf), _p(Pf))
        return dict(xv=xv, Pv=Pv, w=w, xf=xf, Pf=Pf, nf=nf)

    def ekf_state(self, cap=128):
        x = np.zeros(cap, f32)
        P = np.zeros((cap, cap), f32)
        d = self.L.ref_sim_ekf_state(self.h, _p(x), _p(P), cap)
        return x[:d].copy(), P[:d, :d].copy()


class Particles:
    """Oracle particle set (orc_particles*)."""

    def __init__(self, orc, N=None, cap=None, handle=None):
        self.L = orc.lib
        L = self.L
        L.orc_particles_create.restype = C.c_void_p
        L.orc_particles_create.argtypes = [C.c_int, C.c_int]
        L.orc_particles_destroy.argtypes = [C.c_void_p]
        L.orc_particles_n.argtypes = [C.c_void_p]
        L.orc_particles_nf.argtypes = [C.c_void_p]
        L.orc_particles_get.argtypes = [C.c_void_p] * 6
        L.orc_particles_set.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5
        L.orc_estimate.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_predict.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        L.orc_update.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_update_local.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        self.owned = handle is None
        self.h = C.c_void_p(L.orc_particles_create(N, cap)) if handle is None else handle
        self.N = L.orc_particles_n(self.h)

    def nf(self):
        return self.L.orc_particles_nf(self.h)

    def get(self, landmarks=True):
        nf, N = self.nf(), self.N
        xv = np.zeros((N, 3), f32)
        Pv = np.zeros((N, 3, 3), f32)
        w = np.zeros(N, f32)
        xf = np.zeros((N, nf, 2), f32) if landmarks else None
        Pf = np.zeros((N, nf, 2, 2), f32) if landmarks else None
        self.L.orc_particles_get(self.h, _p(xv), _p(Pv), _p(w), _p(xf), _p(Pf))
        return dict(xv=xv, Pv=Pv, w=w, xf=xf, Pf=Pf, nf=nf)

    def weights(self):
        w = np.zeros(self.N, f32)
        self.L.orc_particles_get(self.h, None, None, _p(w), None, None)
        return w

    def set(self, st):
        c = lambda a: np.ascontiguousarray(a, f32)
        self.L.orc_particles_set(self.h, int(st["nf"]), _p(c(st["xv"])), _p(c(st["Pv"])), _p(c(st["w"])),
                                 _p(c(st["xf"])), _p(c(st["Pf"])))

    def estimate(self):
        e = np.zeros(3, np.float64)
        self.L.orc_estimate(self.h, _p(e))
        return e

    def set_log_weights(self, on=True):
        """log-weight extension (slam_oracle.c: orc_particles_set_log_weights): w[] becomes log-weights"""
        self.L.orc_particles_set_log_weights.argtypes = [C.c_void_p, C.c_int]
        self.L.orc_particles_set_log_weights(self.h, int(on))

    def predict(self, algo, V, G, Q, dt, phi_true=0.0, noise2=None):
        self.L.orc_predict(self.h, C.byref(algo), C.c_float(V), C.c_float(G), _p(np.ascontiguousarray(Q, f32)),
                           C.c_float(dt), C.c_float(phi_true), _p(noise2))

    def update(self, algo, zf, idf, zn, R, normals, sel):
        N = self.N
        zf = np.ascontiguousarray(zf, f32).reshape(-1, 2)
        zn = np.ascontiguousarray(zn, f32).reshape(-1, 2)
        idf = np.ascontiguousarray(idf, np.int32)
        keep = np.zeros(N, np.int32)
        neff = C.c_float()
        did = C.c_int()
        self.L.orc_update(self.h, C.byref(algo), _p(zf), _p(idf), zf.shape[0], _p(zn), zn.shape[0],
                          _p(np.ascontiguousarray(R, f32)), _p(normals), _p(sel), _p(keep), C.byref(neff), C.byref(did))
        return keep, f32(neff.value), bool(did.value)

    def update_local(self, algo, zf, idf, zn, R, normals):
        zf = np.ascontiguousarray(zf, f32).reshape(-1, 2)
        zn = np.ascontiguousarray(zn, f32).reshape(-1, 2)
        idf = np.ascontiguousarray(idf, np.int32)
        self.L.orc_update_local(self.h, C.byref(algo), _p(zf), _p(idf), zf.shape[0], _p(zn), zn.shape[0],
                                _p(np.ascontiguousarray(R, f32)), _p(normals))

    def resample_forced(self, algo, sel, forced_did=None, forced_keep=None):
        """resampleParticles alone (orc_resample_forced): forced_did None = the oracle's own decision and ancestors; True / False =
        that decision with forced_keep as the ancestors.  Returns (the oracle's OWN ancestors, its own Neff, its own decision)."""
        L = self.L
        L.orc_resample_forced.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_resample_forced.restype = None
        own = np.zeros(self.N, np.int32)
        neff, did = C.c_float(), C.c_int()
        fk = None if forced_keep is None else np.ascontiguousarray(forced_keep, np.int32)
        L.orc_resample_forced(self.h, C.byref(algo), _p(np.ascontiguousarray(sel, f32)), -1 if forced_did is None else int(bool(forced_did)),
                              _p(fk), _p(own), C.byref(neff), C.byref(did))
        return own, f32(neff.value), bool(did.value)

    def close(self):
        if self.owned and self.h:
            self.L.orc_particles_destroy(self.h)
            self.h = None


class Algo(C.Structure):
    _fields_ = [("method", C.c_int), ("use_heading", C.c_int), ("add_predict_noise", C.c_int), ("resample", C.c_int),
                ("n_effective", C.c_int), ("wheel_base", C.c_float), ("sigma_phi", C.c_float)]


class OrcSim(Sim):
    def __init__(self, orc, args):
        super().__init__(orc.lib, "orc_", args)
        L = self.L
        L.orc_sim_particles.restype = C.c_void_p
        L.orc_sim_particles.argtypes = [C.c_void_p]
        L.orc_sim_set_rng.argtypes = [C.c_void_p, C.c_int, C.c_uint64]
        L.orc_sim_last_resample.argtypes = [C.c_void_p] * 3
        L.orc_sim_last_tape.argtypes = [C.c_void_p] * 3
        L.orc_sim_last_noise2.argtypes = [C.c_void_p] * 2
        L.orc_sim_algo.restype = C.POINTER(Algo)
        L.orc_sim_algo.argtypes = [C.c_void_p]
        L.orc_sim_noise.argtypes = [C.c_void_p] * 4
        L.orc_sim_ekf_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_sim_observe_local.argtypes = [C.c_void_p]
        L.orc_sim_observe_local.restype = None
        L.orc_sim_resample.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_sim_resample.restype = None
        ph = L.orc_sim_particles(self.h)
        self.P = Particles(orc, handle=C.c_void_p(ph)) if ph else None
        self.N = self.P.N if self.P else 0

    def set_rng(self, mode, seed):
        self.L.orc_sim_set_rng(self.h, mode, seed)

    def observe_local(self):
        """observe() without resampleParticles: sensor, association, tape, per-particle update"""
        self.L.orc_sim_observe_local(self.h)

    def resample(self, forced_did=None, forced_keep=None):
        """resampleParticles (core.cpp:718-749).  forced_did None: the oracle's own decision and ancestors; True / False:
        that decision, with forced_keep as the ancestors.  Returns the oracle's OWN ancestors; last_resample() its own
        Neff and decision."""
        own = np.zeros(self.N, np.int32)
        fk = None if forced_keep is None else np.ascontiguousarray(forced_keep, np.int32)
        self.L.orc_sim_resample(self.h, -1 if forced_did is None else int(bool(forced_did)), _p(fk), _p(own))
        return own

    def set_log_weights(self, on=True):
        self.P.set_log_weights(on)

    def nf(self):
        return self.P.nf()

    def estimate(self):
        return self.P.estimate()

    def particles(self, landmarks=True):
        return self.P.get(landmarks)

    def algo(self):
        return self.L.orc_sim_algo(self.h).contents

    def noise(self):
        Q = np.zeros((2, 2), f32)
        R = np.zeros((2, 2), f32)
        dt = C.c_float()
        self.L.orc_sim_noise(self.h, _p(Q), _p(R), C.byref(dt))
        return Q, R, f32(dt.value)

    def last_resample(self):
        ne = C.c_float()
        did = C.c_int()
        self.L.orc_sim_last_resample(self.h, C.byref(ne), C.byref(did))
        return f32(ne.value), bool(did.value)

    def last_tape(self):
        normals = np.zeros((self.N, 3), f32)
        sel = np.zeros(self.N, f32)
        self.L.orc_sim_last_tape(self.h, _p(normals), _p(sel))
        return normals, sel

    def last_noise2(self):
        n2 = np.zeros((self.N, 2), f32)
        self.L.orc_sim_last_noise2(self.h, _p(n2))
        return n2

    def ekf_state(self, cap=128):
        x = np.zeros(cap, f32)
        P = np.zeros((cap, cap), f32)
        d = self.L.orc_sim_ekf_state(self.h, _p(x), _p(P), cap)
        return x[:d].copy(), P[:d, :d].copy()


class Oracle(_FuncLib):
    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        super().__init__(ORACLE_SO, "orc_")
        L = self.lib
        L.orc_llt_lower.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.orc_llt_solve_identity.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.orc_lu_inverse.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.orc_lu_determinant.restype = C.c_float
        L.orc_lu_determinant.argtypes = [C.c_int, C.c_void_p]
        L.orc_srand.argtypes = [C.c_uint]
        L.orc_randn.argtypes = [C.c_int, C.c_int, C.c_void_p]
        L.orc_stratified_random.argtypes = [C.c_int, C.c_void_p]
        L.orc_stratified_resample.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_eigen_sum.restype = C.c_float
        L.orc_eigen_sum.argtypes = [C.c_void_p, C.c_int]
        L.orc_multivariate_gauss.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_fs2_predict_state.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        L.orc_fs1_predict_state.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        L.orc_fs1_compute_weight.restype = C.c_float
        L.orc_fs1_compute_weight.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_void_p]
        L.orc_feature_update.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_void_p]
        L.orc_fs2_sample_proposal.argtypes = [C.c_void_p] * 7 + [C.c_int, C.c_void_p, C.c_void_p]
        L.orc_philox4x32.argtypes = [C.c_uint32] * 6 + [C.c_void_p]
        L.orc_philox_update_tape.argtypes = [C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_philox_predict_tape.argtypes = [C.c_uint64, C.c_uint32, C.c_int, C.c_int, C.c_void_p]
        L.orc_read_map.argtypes = [C.c_char_p, C.POINTER(fp), ip, C.POINTER(fp), ip]
        L.orc_free.argtypes = [C.c_void_p]

    # --- rand() tape ---
    def set_threads(self, n):
        """OpenMP threads of the per-particle loops (bench.py cpu_baseline; results do not depend on it)"""
        self.lib.orc_set_threads.argtypes = [C.c_int]
        self.lib.orc_get_threads.restype = C.c_int
        self.lib.orc_set_threads(int(n))
        return int(self.lib.orc_get_threads())

    def srand(self, seed):
        self.lib.orc_srand(seed)

    def randn(self, m, n):
        out = np.zeros(m * n, f32)
        self.lib.orc_randn(m, n, _p(out))
        return out.reshape(m, n)

    def stratified_random(self, N):
        sel = np.zeros(N, f32)
        cnt = self.lib.orc_stratified_random(N, _p(sel))
        return cnt, sel

    def stratified_resample(self, w, sel):
        N = w.shape[0]
        keep = np.zeros(N, np.int32)
        ne = C.c_float()
        self.lib.orc_stratified_resample(_p(np.ascontiguousarray(w, f32)), N, _p(sel), _p(keep), C.byref(ne))
        return keep, f32(ne.value)

    def eigen_sum(self, v):
        return f32(self.lib.orc_eigen_sum(_p(np.ascontiguousarray(v, f32)), v.shape[0]))

    def multivariate_gauss(self, x, P, g):
        D = x.shape[0]
        out = np.zeros(D, f32)
        self.lib.orc_multivariate_gauss(_p(x), _p(P), D, _p(g), _p(out))
        return out

    def llt_lower(self, A):
        n = A.shape[0]
        L = np.zeros((n, n), f32)
        info = self.lib.orc_llt_lower(n, _p(A), _p(L))
        return L, info

    def llt_solve_identity(self, A):
        n = A.shape[0]
        X = np.zeros((n, n), f32)
        self.lib.orc_llt_solve_identity(n, _p(A), _p(X))
        return X

    def lu_inverse(self, A):
        n = A.shape[0]
        X = np.zeros((n, n), f32)
        self.lib.orc_lu_inverse(n, _p(A), _p(X))
        return X

    def fs2_predict_state(self, xv, Pv, V, G, Q, wb, dt, noise2=None):
        xv = xv.copy()
        Pv = Pv.copy()
        self.lib.orc_fs2_predict_state(_p(xv), _p(Pv), C.c_float(V), C.c_float(G), _p(Q), C.c_float(wb), C.c_float(dt), _p(noise2))
        return xv, Pv

    def fs1_predict_state(self, xv, V, G, Q, wb, dt, noise2=None):
        xv = xv.copy()
        self.lib.orc_fs1_predict_state(_p(xv), C.c_float(V), C.c_float(G), _p(Q), C.c_float(wb), C.c_float(dt), _p(noise2))
        return xv

    def fs1_compute_weight(self, xv, xf, Pf, zf, idf, R):
        return f32(self.lib.orc_fs1_compute_weight(_p(xv), _p(xf), _p(Pf), _p(zf), _p(idf), zf.shape[0], _p(R)))

    def feature_update(self, xv, xf, Pf, zf, idf, R):
        xf = xf.copy()
        Pf = Pf.copy()
        self.lib.orc_feature_update(_p(xv), _p(xf), _p(Pf), _p(zf), _p(idf), zf.shape[0], _p(R))
        return xf, Pf

    def fs2_sample_proposal(self, xv, Pv, w, xf, Pf, zf, idf, R, g3):
        xv = xv.copy()
        Pv = Pv.copy()
        ww = C.c_float(w)
        self.lib.orc_fs2_sample_proposal(_p(xv), _p(Pv), C.byref(ww), _p(xf), _p(Pf), _p(zf), _p(idf), zf.shape[0], _p(R), _p(g3))
        return xv, Pv, f32(ww.value)

    def philox(self, c, k):
        out = np.zeros(4, np.uint32)
        self.lib.orc_philox4x32(*[int(x) for x in c], *[int(x) for x in k], _p(out))
        return out

    def philox_update_tape(self, seed, step, first, count, Ntotal, want_normals=True):
        normals = np.zeros((count, 3), f32) if want_normals else None
        sel = np.zeros(count, f32)
        self.lib.orc_philox_update_tape(seed, step, first, count, Ntotal, _p(normals), _p(sel))
        return normals, sel

    def philox_predict_tape(self, seed, step, first, count):
        n2 = np.zeros((count, 2), f32)
        self.lib.orc_philox_predict_tape(seed, step, first, count, _p(n2))
        return n2

    def read_map(self, path):
        lm = fp()
        wp = fp()
        nlm = C.c_int()
        nwp = C.c_int()
        rc = self.lib.orc_read_map(path.encode(), C.byref(lm), C.byref(nlm), C.byref(wp), C.byref(nwp))
        if rc != 0:
            raise RuntimeError("orc_read_map rc=%d" % rc)
        a = np.ctypeslib.as_array(lm, (2, nlm.value)).copy()
        b = np.ctypeslib.as_array(wp, (2, nwp.value)).copy()
        self.lib.orc_free(lm)
        self.lib.orc_free(wp)
        return a, b

    def sim(self, args):
        return OrcSim(self, args)

    def particles(self, N, cap):
        return Particles(self, N, cap)


class Reference(_FuncLib):
    """The reference's own objects behind oracle/ref_driver.cpp (authoring container only)."""

    def __init__(self, accel=False):
        if accel:
            if not os.path.exists(REF_ACCEL_SO):
                raise RuntimeError("oracle/_ref/libslamref_accel.so not built (needs /root/reference and libslamgpu.so)")
            super().__init__(REF_ACCEL_SO, "ref_")
            if self.lib.ref_accel_init() != 0:
                raise RuntimeError("the reference's AcceleratorHandler (tests/cabi/accel_shim.h) found no GPU")
        else:
            if not have_ref():
                raise RuntimeError("oracle/_ref/libslamref.so not built (needs /root/reference)")
            super().__init__(REF_SO, "ref_")
        L = self.lib
        L.ref_rand_stream.argtypes = [C.c_uint, C.c_int, C.c_void_p]
        L.ref_randn.argtypes = [C.c_uint, C.c_int, C.c_int, C.c_void_p]
        L.ref_multivariate_gauss.argtypes = [C.c_uint, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.ref_stratified_count.argtypes = [C.c_int]
        L.ref_stratified_resample.argtypes = [C.c_uint, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.ref_fs2_predict_state.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_float, C.c_float]
        L.ref_fs1_predict_state.argtypes = [C.c_uint, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_float, C.c_float]
        L.ref_fs1_compute_weight.restype = C.c_float
        L.ref_fs1_compute_weight.argtypes = [C.c_void_p] * 3 + [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.ref_fs2_observe_particle.argtypes = [C.c_uint] + [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.ref_feature_update.argtypes = [C.c_void_p] * 3 + [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        if hasattr(L, "ref_data_associate"):
            L.ref_data_associate.argtypes = [C.c_void_p] * 3 + [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_void_p]

    def data_associate(self, xv, xf, Pf, z, R, gate1, gate2):
        """EKFSLAM::dataAssociate for one particle (pose known): labels [nz]"""
        c = lambda a: np.ascontiguousarray(a, f32)
        xv, xf, Pf, z, R = c(xv), c(xf).reshape(-1, 2), c(Pf).reshape(-1, 2, 2), c(z).reshape(-1, 2), c(R)
        lab = np.zeros(z.shape[0], np.int32)
        self.lib.ref_data_associate(_p(xv), _p(xf), _p(Pf), xf.shape[0], _p(z), z.shape[0], _p(R), gate1, gate2, _p(lab))
        return lab

    def rand_stream(self, seed, count):
        out = np.zeros(count, np.int32)
        self.lib.ref_rand_stream(seed, count, _p(out))
        return out

    def randn(self, seed, m, n):
        out = np.zeros((m, n), f32)
        self.lib.ref_randn(seed, m, n, _p(out))
        return out

    def multivariate_gauss(self, seed, x, P):
        D = x.shape[0]
        out = np.zeros(D, f32)
        self.lib.ref_multivariate_gauss(seed, _p(x), _p(P), D, _p(out))
        return out

    def stratified_count(self, N):
        return self.lib.ref_stratified_count(N)

    def stratified_resample(self, seed, w):
        N = w.shape[0]
        keep = np.zeros(N, np.int32)
        ne = C.c_float()
        self.lib.ref_stratified_resample(seed, _p(np.ascontiguousarray(w, f32)), N, _p(keep), C.byref(ne))
        return keep, f32(ne.value)

    def fs2_predict_state(self, xv, Pv, V, G, Q, wb, dt):
        xv = xv.copy()
        Pv = Pv.copy()
        self.lib.ref_fs2_predict_state(_p(xv), _p(Pv), C.c_float(V), C.c_float(G), _p(Q), C.c_float(wb), C.c_float(dt))
        return xv, Pv

    def fs1_predict_state(self, seed, xv, V, G, Q, wb, dt):
        xv = xv.copy()
        self.lib.ref_fs1_predict_state(seed, _p(xv), C.c_float(V), C.c_float(G), _p(Q), C.c_float(wb), C.c_float(dt))
        return xv

    def fs1_compute_weight(self, xv, xf, Pf, zf, idf, R):
        return f32(self.lib.ref_fs1_compute_weight(_p(xv), _p(xf), _p(Pf), xf.shape[0], _p(zf), _p(idf), zf.shape[0], _p(R)))

    def fs2_observe_particle(self, seed, xv, Pv, w, xf, Pf, zf, idf, R, do_feature_update=True):
        xv = xv.copy()
        Pv = Pv.copy()
        xf = xf.copy()
        Pf = Pf.copy()
        ww = C.c_float(w)
        self.lib.ref_fs2_observe_particle(seed, _p(xv), _p(Pv), C.byref(ww), _p(xf), _p(Pf), xf.shape[0], _p(zf), _p(idf),
                                          zf.shape[0], _p(R), 1 if do_feature_update else 0)
        return xv, Pv, f32(ww.value), xf, Pf

    def feature_update(self, xv, xf, Pf, zf, idf, R):
        xf = xf.copy()
        Pf = Pf.copy()
        self.lib.ref_feature_update(_p(xv), _p(xf), _p(Pf), xf.shape[0], _p(zf), _p(idf), zf.shape[0], _p(R))
        return xf, Pf

    def sim(self, args):
        return RefSim(self, args)
