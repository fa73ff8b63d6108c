"""ctypes binding of include/slamhost.h (libslamhost.so): config / map / vehicle+sensor simulator / known data
association / libc-rand tape — the host-side front end around the hot path."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

DECLARED_SYMBOLS = [
    "slamhost_last_error", "slamhost_sim_create", "slamhost_sim_destroy", "slamhost_sim_conf", "slamhost_sim_map",
    "slamhost_sim_control", "slamhost_sim_observe", "slamhost_sim_last_z", "slamhost_sim_true",
    "slamhost_sim_control_steps", "slamhost_gated_create", "slamhost_gated_destroy", "slamhost_gated_set", "slamhost_gated_step",
    "slamhost_gated_counts", "slamhost_draw_normals", "slamhost_draw_strata", "slamhost_unif_rand",
    "slamhost_synthetic_landmarks", "slamhost_write_map", "slamhost_ekf_create", "slamhost_ekf_destroy", "slamhost_ekf_step",
    "slamhost_ekf_state", "slamhost_plot_open", "slamhost_plot_close", "slamhost_plot_xy", "slamhost_plot_matrix",
    "slamhost_plot_doubles", "slamhost_plot_car_size", "slamhost_plot_u32", "slamhost_plot_cmd", "slamhost_plot_name",
]


class HostConf(C.Structure):
    _fields_ = [(k, C.c_float) for k in ("V", "MAXG", "RATEG", "WHEELBASE", "DT_CONTROLS", "sigmaV", "sigmaG", "MAX_RANGE",
                                         "DT_OBSERVE", "sigmaR", "sigmaB", "sigmaT", "GATE_REJECT", "GATE_AUGMENT", "AT_WAYPOINT")] + \
               [(k, C.c_int32) for k in ("NUMBER_LOOPS", "NPARTICLES", "NEFFECTIVE", "SWITCH_CONTROL_NOISE", "SWITCH_SENSOR_NOISE",
                                         "SWITCH_INFLATE_NOISE", "SWITCH_PREDICT_NOISE", "SWITCH_SAMPLE_PROPOSAL", "SWITCH_HEADING_KNOWN",
                                         "SWITCH_RESAMPLE", "SWITCH_PROFILE", "SWITCH_SEED_RANDOM", "SWITCH_ASSOCIATION_KNOWN",
                                         "SWITCH_BATCH_UPDATE", "SWITCH_USE_IEKF", "method", "n_landmarks", "n_waypoints")] + \
               [("Q", C.c_float * 4), ("R", C.c_float * 4), ("Qe", C.c_float * 4), ("Re", C.c_float * 4)]


def lib_path():
    return os.path.join(_HERE, "libslamhost.so")


_lib = None


def load_library():
    global _lib
    if _lib is None:
        p = lib_path()
        if not os.path.exists(p):
            raise RuntimeError("%s not built: run __graft_entry__.build()" % p)
        L = C.CDLL(p)
        L.slamhost_last_error.restype = C.c_char_p
        L.slamhost_sim_create.restype = C.c_void_p
        L.slamhost_sim_create.argtypes = [C.c_int, C.POINTER(C.c_char_p)]
        L.slamhost_sim_destroy.argtypes = [C.c_void_p]
        L.slamhost_sim_destroy.restype = None
        L.slamhost_sim_conf.argtypes = [C.c_void_p, C.POINTER(HostConf)]
        L.slamhost_sim_map.argtypes = [C.c_void_p] * 3
        L.slamhost_sim_control.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.slamhost_sim_observe.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_void_p, C.POINTER(C.c_int32)]
        L.slamhost_sim_last_z.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
        L.slamhost_sim_true.argtypes = [C.c_void_p, C.c_void_p]
        L.slamhost_sim_true.restype = None
        L.slamhost_sim_control_steps.argtypes = [C.c_void_p]
        L.slamhost_sim_control_steps.restype = C.c_int64
        L.slamhost_draw_normals.argtypes = [C.c_int32, C.c_int32, C.c_void_p]
        L.slamhost_draw_normals.restype = None
        L.slamhost_draw_strata.argtypes = [C.c_int32, C.c_void_p]
        L.slamhost_unif_rand.restype = C.c_double
        L.slamhost_synthetic_landmarks.argtypes = [C.c_uint64, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]
        L.slamhost_synthetic_landmarks.restype = None
        L.slamhost_write_map.argtypes = [C.c_char_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
        L.slamhost_ekf_create.restype = C.c_void_p
        L.slamhost_ekf_create.argtypes = [C.c_void_p]
        L.slamhost_ekf_destroy.argtypes = [C.c_void_p]
        L.slamhost_ekf_destroy.restype = None
        L.slamhost_ekf_step.argtypes = [C.c_void_p, C.c_void_p]
        L.slamhost_ekf_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
        L.slamhost_plot_open.restype = C.c_void_p
        L.slamhost_plot_open.argtypes = [C.c_char_p]
        L.slamhost_plot_close.argtypes = [C.c_void_p]
        L.slamhost_plot_close.restype = None
        L.slamhost_plot_xy.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
        L.slamhost_plot_matrix.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int32]
        L.slamhost_plot_doubles.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int32]
        L.slamhost_plot_car_size.argtypes = [C.c_void_p, C.c_double, C.c_uint32]
        L.slamhost_plot_u32.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32]
        L.slamhost_plot_cmd.argtypes = [C.c_void_p, C.c_char_p]
        L.slamhost_plot_name.argtypes = [C.c_void_p, C.c_char_p]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def draw_normals(count, dim):
    out = np.zeros((count, dim), np.float32)
    load_library().slamhost_draw_normals(count, dim, _p(out))
    return out


def draw_strata(N):
    out = np.zeros(N, np.float32)
    cnt = load_library().slamhost_draw_strata(N, _p(out))
    return cnt, out


def synthetic_landmarks(seed, n, x0, x1, y0, y1):
    lm = np.zeros((2, n), np.float32)
    load_library().slamhost_synthetic_landmarks(seed, n, x0, x1, y0, y1, _p(lm))
    return lm


def write_map(path, lm, wp):
    lm = np.ascontiguousarray(lm, np.float32)
    wp = np.ascontiguousarray(wp, np.float32)
    rc = load_library().slamhost_write_map(path.encode(), _p(lm), lm.shape[1], _p(wp), wp.shape[1])
    if rc != 0:
        raise RuntimeError("slamhost_write_map failed for %s" % path)


class HostSim:
    """The simulator SLAMWrapper::control() + the observation front end (slam-backend CLI arguments)."""

    def __init__(self, args):
        self.L = load_library()
        arr = (C.c_char_p * (len(args) + 1))(b"slam-backend", *[str(a).encode() for a in args])
        self._argv = arr
        h = self.L.slamhost_sim_create(len(args) + 1, arr)
        if not h:
            raise RuntimeError("slamhost_sim_create: %s" % self.L.slamhost_last_error().decode())
        self.h = C.c_void_p(h)
        self.conf = HostConf()
        self.L.slamhost_sim_conf(self.h, C.byref(self.conf))
        self.nlm = self.conf.n_landmarks
        self._zf = np.zeros((self.nlm, 2), np.float32)
        self._zn = np.zeros((self.nlm, 2), np.float32)
        self._idf = np.zeros(self.nlm, np.int32)

    def close(self):
        if self.h:
            self.L.slamhost_sim_destroy(self.h)
            self.h = None

    def map(self):
        lm = np.zeros((2, self.nlm), np.float32)
        wp = np.zeros((2, self.conf.n_waypoints), np.float32)
        self.L.slamhost_sim_map(self.h, _p(lm), _p(wp))
        return lm, wp

    def control(self):
        V, G, phi = C.c_float(), C.c_float(), C.c_float()
        r = self.L.slamhost_sim_control(self.h, C.byref(V), C.byref(G), C.byref(phi))
        return r, V.value, G.value, phi.value

    def observe(self, nf_known):
        m, n = C.c_int32(), C.c_int32()
        self.L.slamhost_sim_observe(self.h, nf_known, _p(self._zf), _p(self._idf), C.byref(m), _p(self._zn), C.byref(n))
        return self._zf[:m.value].copy(), self._idf[:m.value].copy(), self._zn[:n.value].copy()

    def last_z(self):
        """raw observation of the last observe(): z [nz, 2] (range, bearing), visible landmark ids [nz]"""
        z = np.zeros((self.nlm, 2), np.float32)
        vis = np.zeros(self.nlm, np.int32)
        nz = C.c_int32()
        self.L.slamhost_sim_last_z(self.h, _p(z), _p(vis), C.byref(nz))
        return z[:nz.value].copy(), vis[:nz.value].copy()

    def true_pose(self):
        x = np.zeros(3, np.float32)
        self.L.slamhost_sim_true(self.h, _p(x))
        return x

    def noise(self):
        f = lambda a: np.array(list(a), np.float32).reshape(2, 2)
        return f(self.conf.Qe), f(self.conf.Re), np.float32(self.conf.DT_CONTROLS)


class GatedPolicy:
    """the policy between the gated association's vote and the update's packet (include/slamhost.h: slamhost_gated_*;
    slam_amd/csrc/host/gated.h): what slam-backend -assoc gated applies every step"""

    def __init__(self, **tunables):
        self.L = load_library()
        L = self.L
        L.slamhost_gated_create.restype = C.c_void_p
        L.slamhost_gated_destroy.argtypes = [C.c_void_p]
        L.slamhost_gated_set.argtypes = [C.c_void_p, C.c_char_p, C.c_double]
        L.slamhost_gated_step.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_int32,
                                          C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_void_p, C.POINTER(C.c_int32), C.c_void_p, C.POINTER(C.c_int32)]
        L.slamhost_gated_counts.argtypes = [C.c_void_p, C.c_void_p]
        self.h = C.c_void_p(L.slamhost_gated_create())
        for k, v in tunables.items():
            if L.slamhost_gated_set(self.h, k.encode(), float(v)) != 0:
                raise ValueError("unknown association-policy tunable %r" % k)

    def step(self, z, consensus, support, xv, xf, max_range, room):
        """-> (zf [m, 2], idf [m], zn [n, 2], retire [r]) for slamgpu_update / slamgpu_retire_landmarks"""
        z = np.ascontiguousarray(z, np.float32).reshape(-1, 2)
        nz = z.shape[0]
        xf = np.ascontiguousarray(xf, np.float32).reshape(-1, 2)
        nf = xf.shape[0]
        cons = np.ascontiguousarray(consensus, np.int32)
        sup = np.ascontiguousarray(support, np.float32)
        xv = np.ascontiguousarray(xv, np.float32)
        zf, idf, zn, ret = np.zeros((max(nz, 1), 2), np.float32), np.zeros(max(nz, 1), np.int32), np.zeros((max(nz, 1), 2), np.float32), np.zeros(max(nf, 1), np.int32)
        m, n, r = C.c_int32(), C.c_int32(), C.c_int32()
        rc = self.L.slamhost_gated_step(self.h, _p(z), nz, _p(cons), _p(sup), _p(xv), _p(xf), nf, float(max_range), int(room), _p(zf), _p(idf), C.byref(m), _p(zn),
                                        C.byref(n), _p(ret), C.byref(r))
        if rc != 0:
            raise RuntimeError("slamhost_gated_step: bad arguments")
        return zf[:m.value].copy(), idf[:m.value].copy(), zn[:n.value].copy(), ret[:r.value].copy()

    def counts(self):
        c = np.zeros(6, np.int32)
        self.L.slamhost_gated_counts(self.h, _p(c))
        return dict(zip(("opened", "retired", "second_stage_matches", "unused", "refused_new", "in_use"), (int(x) for x in c)))

    def close(self):
        if self.h:
            self.L.slamhost_gated_destroy(self.h)
            self.h = None


class HostEkf:
    """EKF-SLAM on the host CPU (-method EKF1): EKFSLAMWrapper's loop body over a HostSim."""

    def __init__(self, sim):
        self.L, self.sim = sim.L, sim
        self.h = C.c_void_p(self.L.slamhost_ekf_create(sim.h))

    def step(self):
        return self.L.slamhost_ekf_step(self.h, self.sim.h)

    def state(self, cap=128, want_P=True):
        x = np.zeros(cap, np.float32)
        P = np.zeros((cap, cap), np.float32) if want_P else None
        d = self.L.slamhost_ekf_state(self.h, _p(x), _p(P), cap)
        return x[:d].copy(), (P[:d, :d].copy() if want_P else None)

    def close(self):
        if self.h:
            self.L.slamhost_ekf_destroy(self.h)
            self.h = None


def make_tape(args, max_obs=None):
    """Whole control / observation tape of a run (RNG: libc rand() seeded by SWITCH_SEED_RANDOM for the control and
    sensor noise only — the particle noise is not drawn here).  Returns a list of observation steps, each with the
    controls of the predicts that precede it."""
    sim = HostSim(args)
    steps, ctl, nf = [], [], 0
    while True:
        r, V, G, phi = sim.control()
        if r < 0:
            break
        ctl.append((V, G, phi))
        if r == 1:
            zf, idf, zn = sim.observe(nf)
            steps.append(dict(controls=ctl, zf=zf, idf=idf, zn=zn, nf_before=nf, true=sim.true_pose()))
            nf += zn.shape[0]
            ctl = []
            if max_obs and len(steps) >= max_obs:
                break
    Q, R, dt = sim.noise()
    conf = sim.conf
    sim.close()
    return dict(steps=steps, Q=Q, R=R, dt=dt, conf=conf, nlm=conf.n_landmarks, tail_controls=ctl)


class Plot:
    """NetworkPlot's method surface (src/backend/plotting/NetworkPlot.cpp) over the product's wire encoder and sinks
    (include/slamhost.h: slamhost_plot_*).  spec: tcp://host:port | file:<path> | gather:<dir> | none, comma-separated."""

    def __init__(self, spec):
        self.L = load_library()
        h = self.L.slamhost_plot_open(spec.encode())
        if not h:
            raise RuntimeError("slamhost_plot_open: %s" % self.L.slamhost_last_error().decode())
        self.h = C.c_void_p(h)

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError("plot: %s" % self.L.slamhost_last_error().decode())

    def close(self):
        if self.h:
            self.L.slamhost_plot_close(self.h)
            self.h = None

    def xy(self, cmd, xs, ys):
        xs, ys = np.ascontiguousarray(xs, np.float64), np.ascontiguousarray(ys, np.float64)
        self._chk(self.L.slamhost_plot_xy(self.h, cmd.encode(), _p(xs), xs.size, _p(ys), ys.size))

    def matrix(self, cmd, a, idx=0):
        a = np.ascontiguousarray(a, np.float32)
        rows, cols = (a.shape if a.ndim == 2 else (0, 0))
        self._chk(self.L.slamhost_plot_matrix(self.h, cmd.encode(), rows, cols, _p(a), idx))

    def doubles(self, cmd, *v):
        a = np.array(v, np.float64)
        self._chk(self.L.slamhost_plot_doubles(self.h, cmd.encode(), _p(a), a.size))

    def car_size(self, s, ident):
        self._chk(self.L.slamhost_plot_car_size(self.h, s, ident))

    def u32(self, cmd, v):
        self._chk(self.L.slamhost_plot_u32(self.h, cmd.encode(), v))

    def cmd(self, cmd):
        self._chk(self.L.slamhost_plot_cmd(self.h, cmd.encode()))

    def name(self, n):
        self._chk(self.L.slamhost_plot_name(self.h, n.encode()))
