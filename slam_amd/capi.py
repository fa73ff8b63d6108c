"""ctypes binding of include/slamgpu.h (the C ABI of libslamgpu.so).  Fails loudly when the library is
missing: there is deliberately no fallback path."""
import ctypes as C
import os

import numpy as np

FASTSLAM1, FASTSLAM2 = 1, 2
RNG_TAPE, RNG_PHILOX = 0, 1
MATH_STRICT, MATH_FAST = 0, 1
ASSOC_NEW, ASSOC_DISCARD = -1, -2

_HERE = os.path.dirname(os.path.abspath(__file__))

# every symbol include/slamgpu.h declares (tests check the built library exports each one)
DECLARED_SYMBOLS = [
    "slamgpu_last_error", "slamgpu_abi_version", "slamgpu_device_count", "slamgpu_jacobians", "slamgpu_jacobians_multi", "slamgpu_create",
    "slamgpu_destroy", "slamgpu_predict", "slamgpu_update", "slamgpu_estimate", "slamgpu_estimate_async", "slamgpu_estimate_fetch", "slamgpu_stats", "slamgpu_ancestors",
    "slamgpu_num_landmarks", "slamgpu_retire_landmarks", "slamgpu_genealogy_rows", "slamgpu_persist_info", "slamgpu_persist_status", "slamgpu_download", "slamgpu_upload", "slamgpu_sync", "slamgpu_step", "slamgpu_history_fetch", "slamgpu_shard_set_totals_buffer", "slamgpu_shard_step", "slamgpu_timer_start", "slamgpu_timer_stop", "slamgpu_stream", "slamgpu_profile",
    "slamgpu_kernel_time", "slamgpu_algorithmic_bytes", "slamgpu_shard_update", "slamgpu_shard_block_totals", "slamgpu_shard_plan",
    "slamgpu_shard_record_floats", "slamgpu_shard_pack", "slamgpu_shard_unpack", "slamgpu_shard_finish", "slamgpu_shard_estimate",
    "slamgpu_dev_alloc", "slamgpu_dev_free", "slamgpu_dev_copy", "slamgpu_dev_copy_async", "slamgpu_shard_estimate_async",
    "slamgpu_shard_estimate_fetch", "slamgpu_step_status", "slamgpu_kat", "slamgpu_download_range", "slamgpu_debug_stamps", "slamgpu_associate", "slamgpu_set_map", "slamgpu_observe",
    "slamgpu_dist_export_size", "slamgpu_dist_export", "slamgpu_dist_connect", "slamgpu_dist_step", "slamgpu_dist_totals", "slamgpu_dist_settle",
    "slamgpu_dist_history_fetch", "slamgpu_dist_gather", "slamgpu_dist_set_collective", "slamgpu_dist_handshake_test",
    "slamgpu_dist_collective_status", "slamgpu_dist_comm_id", "slamgpu_dist_comm_init", "slamgpu_dist_group_create", "slamgpu_dist_group_destroy",
    "slamgpu_dist_group_step", "slamgpu_dist_group_settle", "slamgpu_dist_group_history", "slamgpu_dist_group_download",
    "slamgpu_peek", "slamgpu_step_observe", "slamgpu_run_observe", "slamgpu_observe_fetch", "slamgpu_associate_ex", "slamgpu_update_particle", "slamgpu_update_labels", "slamgpu_dist_comm_info", "slamgpu_dist_remote_reads",
]
ASSOC_AUTO, ASSOC_EXHAUSTIVE, ASSOC_GRID = 0, 1, 2
FLAG_DEVICE_OBSERVE = 1
FLAG_NO_REFERENCE_RESAMPLE = 2
FLAG_PARTICLE_MAPS = 4


class SlamGpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("slamgpu error %d: %s" % (code, msg))
        self.code = code


class ParticleAssoc(C.Structure):  # slamgpu_particle_assoc
    _fields_ = [("gate_reject", C.c_float), ("gate_augment", C.c_float), ("mode", C.c_int32), ("new_share", C.c_float), ("p_new", C.c_float),
                ("census_every", C.c_int32), ("excl_base", C.c_float), ("excl_per_m", C.c_float), ("unique_ratio", C.c_float)]


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("method", C.c_int32), ("n_particles", C.c_int32),
                ("max_landmarks", C.c_int32), ("use_heading", C.c_int32), ("add_predict_noise", C.c_int32),
                ("resample", C.c_int32), ("n_effective", C.c_int32), ("wheel_base", C.c_float), ("sigma_phi", C.c_float),
                ("rng_mode", C.c_int32), ("math_mode", C.c_int32), ("seed", C.c_uint64), ("first_particle", C.c_int64),
                ("n_particles_global", C.c_int64), ("external_stream", C.c_uint64), ("log_weights", C.c_int32),
                ("flags", C.c_int32)]


class ShardPlan(C.Structure):
    """slamgpu_shard_plan_t"""
    _fields_ = [("wsum", C.c_double), ("wsq", C.c_double), ("neff", C.c_float), ("resampled", C.c_int32), ("K", C.c_int64 * 65),
                ("status", C.c_int32), ("pad", C.c_int32)]


def lib_path():
    # SLAMGPU_LIB: diagnostic override (tools/stamps.py loads the instrumented libslamgpu_stamps.so)
    return os.environ.get("SLAMGPU_LIB") or os.path.join(_HERE, "libslamgpu.so")


_lib = None


def load_library():
    """Load libslamgpu.so (built in-tree by __graft_entry__.build() / slam_amd/csrc/Makefile)."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not os.path.exists(p):
        raise SlamGpuError(-4, "%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(no CPU fallback exists)" % p)
    L = C.CDLL(p)
    L.slamgpu_last_error.restype = C.c_char_p
    L.slamgpu_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
    L.slamgpu_destroy.argtypes = [C.c_void_p]
    L.slamgpu_destroy.restype = None
    L.slamgpu_jacobians.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    L.slamgpu_jacobians_multi.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64]
    L.slamgpu_predict.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_float, C.c_float, C.c_void_p]
    L.slamgpu_update.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                 C.c_void_p, C.c_void_p]
    L.slamgpu_step.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int32,
                               C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    L.slamgpu_shard_step.argtypes = L.slamgpu_step.argtypes[:-1]
    L.slamgpu_estimate.argtypes = [C.c_void_p, C.c_void_p]
    L.slamgpu_estimate_async.argtypes = [C.c_void_p]
    L.slamgpu_estimate_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
    L.slamgpu_history_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
    L.slamgpu_step_status.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    L.slamgpu_kat.argtypes = [C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
    L.slamgpu_associate.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    L.slamgpu_associate_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_float, C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p]
    L.slamgpu_set_map.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    L.slamgpu_observe.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int32] + [C.c_void_p] * 2 + [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_void_p, C.POINTER(C.c_int32)]
    L.slamgpu_step_observe.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_float, C.c_void_p, C.c_float, C.c_void_p, C.c_int32,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    L.slamgpu_run_observe.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_float, C.c_void_p, C.c_int32]
    L.slamgpu_observe_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_void_p, C.c_void_p, C.POINTER(C.c_int32),
                                        C.c_void_p, C.POINTER(C.c_int32)]
    L.slamgpu_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
    L.slamgpu_stats.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    L.slamgpu_ancestors.argtypes = [C.c_void_p, C.c_void_p]
    L.slamgpu_num_landmarks.argtypes = [C.c_void_p]
    L.slamgpu_update_particle.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.slamgpu_update_labels.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    if hasattr(L, "slamgpu_retire_landmarks"):
        L.slamgpu_retire_landmarks.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    L.slamgpu_genealogy_rows.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.slamgpu_persist_info.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
    if hasattr(L, "slamgpu_persist_status"):  # (an older build loaded through SLAMGPU_LIB for an A/B lacks the round-6 entries)
        L.slamgpu_persist_status.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.slamgpu_download.argtypes = [C.c_void_p] * 6
    L.slamgpu_download_range.argtypes = [C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 5
    L.slamgpu_peek.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 5
    L.slamgpu_upload.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 5
    L.slamgpu_sync.argtypes = [C.c_void_p]
    L.slamgpu_stream.argtypes = [C.c_void_p]
    L.slamgpu_stream.restype = C.c_void_p
    L.slamgpu_timer_start.argtypes = [C.c_void_p]
    L.slamgpu_timer_stop.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    L.slamgpu_profile.argtypes = [C.c_void_p, C.c_int32]
    L.slamgpu_kernel_time.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.slamgpu_algorithmic_bytes.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.slamgpu_shard_update.argtypes = L.slamgpu_update.argtypes
    L.slamgpu_shard_set_totals_buffer.argtypes = [C.c_void_p, C.c_void_p]
    L.slamgpu_shard_block_totals.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int32)]
    L.slamgpu_shard_plan.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(ShardPlan)]
    L.slamgpu_shard_record_floats.argtypes = [C.c_void_p]
    L.slamgpu_shard_pack.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(ShardPlan), C.c_void_p,
                                     C.c_void_p, C.c_void_p]
    L.slamgpu_shard_unpack.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(ShardPlan)]
    L.slamgpu_shard_finish.argtypes = [C.c_void_p, C.POINTER(ShardPlan)]
    L.slamgpu_shard_estimate.argtypes = [C.c_void_p, C.c_void_p]
    L.slamgpu_dev_alloc.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_void_p)]
    L.slamgpu_dev_free.argtypes = [C.c_void_p, C.c_void_p]
    L.slamgpu_dev_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    L.slamgpu_dev_copy_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    L.slamgpu_shard_estimate_async.argtypes = [C.c_void_p]
    L.slamgpu_shard_estimate_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
    L.slamgpu_dist_export_size.argtypes = []
    L.slamgpu_dist_export.argtypes = [C.c_void_p, C.c_void_p]
    L.slamgpu_dist_connect.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    L.slamgpu_dist_step.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int32,
                                    C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
    L.slamgpu_dist_totals.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int32)]
    L.slamgpu_dist_settle.argtypes = [C.c_void_p]
    L.slamgpu_dist_history_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
    L.slamgpu_dist_gather.argtypes = [C.c_void_p]
    L.slamgpu_dist_set_collective.argtypes = [C.c_void_p, C.c_int32]
    L.slamgpu_dist_handshake_test.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    L.slamgpu_dist_collective_status.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    L.slamgpu_dist_comm_id.argtypes = [C.c_void_p, C.c_int32]
    L.slamgpu_dist_comm_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
    L.slamgpu_dist_group_create.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(C.c_void_p)]
    L.slamgpu_dist_group_destroy.argtypes = [C.c_void_p]
    L.slamgpu_dist_group_destroy.restype = None
    L.slamgpu_dist_group_step.argtypes = [C.c_void_p] + L.slamgpu_dist_step.argtypes[1:]
    L.slamgpu_dist_group_settle.argtypes = [C.c_void_p]
    L.slamgpu_dist_group_history.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
    L.slamgpu_dist_group_download.argtypes = [C.c_void_p] * 6
    _lib = L
    return L


def _chk(rc):
    if rc != 0:
        raise SlamGpuError(rc, load_library().slamgpu_last_error().decode())


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, np.float32)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def device_count():
    return load_library().slamgpu_device_count()


def jacobians(xv, R, xf, Pf):
    """Seam 1 (AcceleratorHandler window, core.cpp:586-664).  R, Pf given row-major here; packed column-major."""
    n = xf.shape[0]
    buf = np.zeros(7 + 6 * n, np.float32)
    buf[:3] = xv
    buf[3:7] = np.asarray(R, np.float32).T.ravel()  # Eigen linear index = column-major
    rec = buf[7:].reshape(n, 6)
    rec[:, :2] = xf
    rec[:, 2:] = np.asarray(Pf, np.float32).transpose(0, 2, 1).reshape(n, 4)
    out = np.zeros(16 * n, np.float32)
    _chk(load_library().slamgpu_jacobians(_ptr(buf), n, _ptr(out)))
    o = out.reshape(n, 16)
    return o[:, 0:2].copy(), o[:, 6:12].reshape(n, 2, 3).copy(), o[:, 2:6].reshape(n, 2, 2).copy(), o[:, 12:16].reshape(n, 2, 2).copy()


def kat(math_mode, op, data):
    """slamgpu_kat: op 0 trig_offset [n], 1 gaussEvaluate D=2 [n,5], 2 gaussEvaluate D=3 [n,9] -> [n]"""
    a = _f32(data)
    per = {0: 1, 1: 5, 2: 9}.get(op, 1)
    n = a.size // per
    out = np.zeros(n, np.float32)
    _chk(load_library().slamgpu_kat(math_mode, op, _ptr(a), n, _ptr(out)))
    return out


def dist_comm_id():
    """a fresh RCCL unique id (rank 0 makes it, every rank passes it to SlamGpu.dist_comm_init)"""
    buf = C.create_string_buffer(128)
    _chk(load_library().slamgpu_dist_comm_id(buf, 128))
    return buf.raw


class DistGroup:
    """All shards of a distributed run in this process (slamgpu_dist_group_*): k contexts on k GPUs (RCCL inside the
    library) or k logical shards on one GPU sharing `stream`."""

    def __init__(self, n_shards, n_per_shard, max_landmarks, devices=None, stream=0, **kw):
        self.L = load_library()
        kw.setdefault("rng_mode", RNG_PHILOX)
        self.G, self.n = n_shards, n_per_shard
        self.ctx = []
        for g in range(n_shards):
            dev = devices[g] if devices else 0
            ext = stream
            if not devices and n_shards > 1 and not stream and self.ctx:
                ext = self.ctx[0].stream()  # logical shards: everybody on the first context's stream
            self.ctx.append(SlamGpu(n_per_shard, max_landmarks, first_particle=g * n_per_shard, n_particles_global=n_shards * n_per_shard,
                                    device=dev, external_stream=ext or 0, **kw))
        arr = (C.c_void_p * n_shards)(*[c.h for c in self.ctx])
        self.h = C.c_void_p()
        _chk(self.L.slamgpu_dist_group_create(arr, n_shards, C.byref(self.h)))

    def prepare_step(self, controls, Q, dt, zf, idf, zn, R, record_estimate=True):
        ctl = _f32(controls).reshape(-1, 3)
        Q = _f32(Q, 4)
        zf = _f32(zf).reshape(-1, 2)
        zn = _f32(zn).reshape(-1, 2)
        idf = np.ascontiguousarray(idf, np.int32)
        R = _f32(R, 4)
        keep = (ctl, Q, zf, zn, idf, R)
        args = (self.h, _ptr(ctl), ctl.shape[0], _ptr(Q), C.c_float(dt), _ptr(zf), _ptr(idf), zf.shape[0], _ptr(zn), zn.shape[0],
                _ptr(R), 1 if record_estimate else 0)
        fn = self.L.slamgpu_dist_group_step

        def call(_keep=keep):
            _chk(fn(*args))
        return call

    def step(self, controls, Q, dt, zf, idf, zn, R, record_estimate=True):
        self.prepare_step(controls, Q, dt, zf, idf, zn, R, record_estimate)()

    def settle(self):
        _chk(self.L.slamgpu_dist_group_settle(self.h))

    def history_fetch(self, max_count=4096):
        out = np.zeros((max_count, 3), np.float64)
        ne = np.zeros(max_count, np.float32)
        rs = np.zeros(max_count, np.int32)
        st = np.zeros(max_count, np.int32)
        n = C.c_int32()
        _chk(self.L.slamgpu_dist_group_history(self.h, _ptr(out), _ptr(ne), _ptr(rs), _ptr(st), max_count, C.byref(n)))
        self.last_history_status = st[:n.value].copy()
        return out[:n.value].copy(), ne[:n.value].copy(), rs[:n.value].astype(bool)

    def download(self, landmarks=True):
        N, nf = self.G * self.n, self.ctx[0].nf()
        xv = np.zeros((N, 3), np.float32)
        Pv = np.zeros((N, 3, 3), np.float32)
        w = np.zeros(N, np.float32)
        xf = np.zeros((N, nf, 2), np.float32) if landmarks else None
        Pf = np.zeros((N, nf, 2, 2), np.float32) if landmarks else None
        _chk(self.L.slamgpu_dist_group_download(self.h, _ptr(xv), _ptr(Pv), _ptr(w), _ptr(xf), _ptr(Pf)))
        return dict(nf=nf, xv=xv, Pv=Pv, w=w, xf=xf, Pf=Pf)

    def sync(self):
        for c in self.ctx:
            _chk(self.L.slamgpu_sync(c.h))

    def close(self):
        if self.h:
            self.L.slamgpu_dist_group_destroy(self.h)
            self.h = C.c_void_p()
        for c in self.ctx:
            c.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class SlamGpu:
    """One device-resident particle set = one FastSLAM{1,2} algorithm object (fastslam2.h:20-31)."""

    def __init__(self, n_particles, max_landmarks, method=FASTSLAM2, n_effective=None, resample=True, use_heading=False,
                 add_predict_noise=None, wheel_base=4.0, sigma_phi=0.017453292519943, rng_mode=RNG_TAPE, seed=0,
                 math_mode=MATH_STRICT, device=0, first_particle=0, n_particles_global=0, external_stream=0, log_weights=False,
                 device_observe=False, reference_resample=True, particle_maps=False):
        self.L = load_library()
        cfg = Config()
        cfg.struct_size = C.sizeof(Config)
        cfg.device = device
        cfg.method = method
        cfg.n_particles = n_particles
        cfg.max_landmarks = max_landmarks
        cfg.use_heading = int(use_heading)
        cfg.add_predict_noise = int(method == FASTSLAM1 if add_predict_noise is None else add_predict_noise)
        cfg.resample = int(resample)
        ng = n_particles_global or n_particles
        cfg.n_effective = int(0.75 * ng) if n_effective is None else int(n_effective)
        cfg.wheel_base = wheel_base
        cfg.sigma_phi = sigma_phi
        cfg.rng_mode = rng_mode
        cfg.math_mode = math_mode
        cfg.seed = seed
        cfg.first_particle = first_particle
        cfg.n_particles_global = ng
        cfg.external_stream = external_stream
        cfg.log_weights = int(log_weights)
        cfg.flags = ((FLAG_DEVICE_OBSERVE if device_observe else 0) | (0 if reference_resample else FLAG_NO_REFERENCE_RESAMPLE) |
                     (FLAG_PARTICLE_MAPS if particle_maps else 0))
        self.cfg = cfg
        self.N = n_particles
        self.h = C.c_void_p()
        _chk(self.L.slamgpu_create(C.byref(cfg), C.byref(self.h)))

    def close(self):
        if self.h:
            self.L.slamgpu_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def predict(self, V, G, Q, dt, phi_true=0.0, noise2=None):
        Q = _f32(Q, 4)
        n2 = None if noise2 is None else _f32(noise2, (self.N, 2))
        _chk(self.L.slamgpu_predict(self.h, V, G, _ptr(Q), dt, phi_true, _ptr(n2)))

    def update(self, zf, idf, zn, R, normals=None, strata=None):
        zf = _f32(zf).reshape(-1, 2)
        zn = _f32(zn).reshape(-1, 2)
        idf = np.ascontiguousarray(idf, np.int32)
        R = _f32(R, 4)
        nm = None if normals is None else _f32(normals, (self.N, 3))
        st = None if strata is None else _f32(strata)
        _chk(self.L.slamgpu_update(self.h, _ptr(zf), _ptr(idf), zf.shape[0], _ptr(zn), zn.shape[0], _ptr(R), _ptr(nm), _ptr(st)))

    def prepare_step(self, controls, Q, dt, zf, idf, zn, R, normals=None, strata=None, record_estimate=True, shard=False):
        """Marshal one filter step (k predicts + update [+ estimate_async]) once; the returned callable makes the single
        slamgpu_step call.  Lets a driver loop pay the numpy -> pointer conversions outside its timed region, the way a
        C++ host that already holds plain arrays would."""
        ctl = _f32(controls).reshape(-1, 3)
        Q = _f32(Q, 4)
        zf = _f32(zf).reshape(-1, 2)
        zn = _f32(zn).reshape(-1, 2)
        idf = np.ascontiguousarray(idf, np.int32)
        R = _f32(R, 4)
        nm = None if normals is None else _f32(normals, (self.N, 3))
        st = None if strata is None else _f32(strata)
        keep = (ctl, Q, zf, zn, idf, R, nm, st)  # the pointers below borrow these buffers
        args = (self.h, _ptr(ctl), ctl.shape[0], _ptr(Q), C.c_float(dt), _ptr(zf), _ptr(idf), zf.shape[0], _ptr(zn), zn.shape[0],
                _ptr(R), _ptr(nm), _ptr(st))
        if shard:
            fn = self.L.slamgpu_shard_step
        else:
            fn = self.L.slamgpu_step
            args = args + (1 if record_estimate else 0,)

        def call(_keep=keep):
            _chk(fn(*args))
        return call

    def step(self, controls, Q, dt, zf, idf, zn, R, normals=None, strata=None, record_estimate=True):
        self.prepare_step(controls, Q, dt, zf, idf, zn, R, normals, strata, record_estimate)()

    def shard_step(self, controls, Q, dt, zf, idf, zn, R, normals=None, strata=None):
        self.prepare_step(controls, Q, dt, zf, idf, zn, R, normals, strata, shard=True)()

    def estimate(self):
        e = np.zeros(3, np.float64)
        _chk(self.L.slamgpu_estimate(self.h, _ptr(e)))
        return e

    def estimate_async(self):
        _chk(self.L.slamgpu_estimate_async(self.h))

    def estimate_fetch(self, max_count=4096):
        out = np.zeros((max_count, 3), np.float64)
        n = C.c_int32()
        _chk(self.L.slamgpu_estimate_fetch(self.h, _ptr(out), max_count, C.byref(n)))
        return out[:n.value].copy()

    def history_fetch(self, max_count=4096):
        """(xyt[k,3], neff[k], resampled[k]) of the estimate_async entries recorded since the last fetch."""
        out = np.zeros((max_count, 3), np.float64)
        ne = np.zeros(max_count, np.float32)
        rs = np.zeros(max_count, np.int32)
        n = C.c_int32()
        st = np.zeros(max_count, np.int32)
        _chk(self.L.slamgpu_history_fetch(self.h, _ptr(out), _ptr(ne), _ptr(rs), _ptr(st), max_count, C.byref(n)))
        self.last_history_status = st[:n.value].copy()
        return out[:n.value].copy(), ne[:n.value].copy(), rs[:n.value].astype(bool)

    def status(self):
        """SLAMGPU_STATUS_* bits of the last update's resampling stage"""
        st = C.c_int32()
        _chk(self.L.slamgpu_step_status(self.h, C.byref(st)))
        return st.value

    def stats(self):
        ne, rs, ws = C.c_float(), C.c_int32(), C.c_double()
        _chk(self.L.slamgpu_stats(self.h, C.byref(ne), C.byref(rs), C.byref(ws)))
        return np.float32(ne.value), bool(rs.value), ws.value

    def ancestors(self):
        keep = np.zeros(self.N, np.int32)
        _chk(self.L.slamgpu_ancestors(self.h, _ptr(keep)))
        return keep

    def nf(self):
        n = self.L.slamgpu_num_landmarks(self.h)
        if n < 0:  # a negative slamgpu_status (e.g. SLAMGPU_ERR_CAPACITY from the device front end), never a count
            _chk(n)
        return n

    def retire_landmarks(self, ids):
        """landmarks that take no further part in the gated association (slamgpu_retire_landmarks)"""
        ids = np.ascontiguousarray(ids, np.int32)
        _chk(self.L.slamgpu_retire_landmarks(self.h, _ptr(ids), int(ids.size)))

    def live_rows(self):
        """genealogy rows in use (slamgpu_genealogy_rows)"""
        a, b = C.c_int32(), C.c_int32()
        _chk(self.L.slamgpu_genealogy_rows(self.h, C.byref(a), C.byref(b)))
        return a.value

    def genealogy_rows(self):
        """(rows in use, row capacity) (slamgpu_genealogy_rows): capacity 40 = the compact layout"""
        a, b = C.c_int32(), C.c_int32()
        _chk(self.L.slamgpu_genealogy_rows(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def persist_info(self, cross=False):
        """(launches, iterations[, cross_xcd]) of the persistent step loop behind run_observe (slamgpu_persist_info)"""
        a, b, x = C.c_int64(), C.c_int64(), C.c_int32()
        _chk(self.L.slamgpu_persist_info(self.h, C.byref(a), C.byref(b), C.byref(x) if cross else None))
        return (a.value, b.value, x.value) if cross else (a.value, b.value)

    def persist_status(self):
        """None, or (launch, completed, handed) of the abandoned launch of the persistent step loop (slamgpu_persist_status)"""
        ab, la, co, ha = C.c_int32(), C.c_int64(), C.c_int32(), C.c_int32()
        _chk(self.L.slamgpu_persist_status(self.h, C.byref(ab), C.byref(la), C.byref(co), C.byref(ha)))
        return (la.value, co.value, ha.value) if ab.value else None

    def download(self, landmarks=True, first=0, count=None):
        """particles [first, first + count) (default: all); log-weight contexts: w holds log-weights"""
        nf = self.nf()
        N = self.N - first if count is None else count
        xv = np.zeros((N, 3), np.float32)
        Pv = np.zeros((N, 3, 3), np.float32)
        w = np.zeros(N, np.float32)
        xf = np.zeros((N, nf, 2), np.float32) if landmarks else None
        Pf = np.zeros((N, nf, 2, 2), np.float32) if landmarks else None
        _chk(self.L.slamgpu_download_range(self.h, first, N, _ptr(xv), _ptr(Pv), _ptr(w), _ptr(xf), _ptr(Pf)))
        return dict(xv=xv, Pv=Pv, w=w, xf=xf, Pf=Pf, nf=nf)

    def peek(self, landmarks=True, first=0, stride=1, count=None):
        """slamgpu_peek: the same dict as download() for particles first, first + stride, ..., without rewriting any state
        (pose through a pending gather, records through the genealogy)"""
        nf = self.nf()
        N = (self.N - first + stride - 1) // stride if count is None else count
        xv = np.zeros((N, 3), np.float32)
        Pv = np.zeros((N, 3, 3), np.float32)
        w = np.zeros(N, np.float32)
        xf = np.zeros((N, nf, 2), np.float32) if landmarks else None
        Pf = np.zeros((N, nf, 2, 2), np.float32) if landmarks else None
        _chk(self.L.slamgpu_peek(self.h, first, stride, N, _ptr(xv), _ptr(Pv), _ptr(w), _ptr(xf), _ptr(Pf)))
        return dict(xv=xv, Pv=Pv, w=w, xf=xf, Pf=Pf, nf=nf)

    def upload(self, st):
        nf = int(st["nf"])
        xf = _f32(st["xf"]) if nf else None
        Pf = _f32(st["Pf"]) if nf else None
        _chk(self.L.slamgpu_upload(self.h, nf, _ptr(_f32(st["xv"])), _ptr(_f32(st["Pv"])), _ptr(_f32(st["w"])), _ptr(xf), _ptr(Pf)))

    def sync(self):
        _chk(self.L.slamgpu_sync(self.h))

    # ---- sharded operation (include/slamgpu.h "sharded operation") ----
    def shard_update(self, zf, idf, zn, R, normals=None, strata=None):
        zf = _f32(zf).reshape(-1, 2)
        zn = _f32(zn).reshape(-1, 2)
        idf = np.ascontiguousarray(idf, np.int32)
        R = _f32(R, 4)
        nm = None if normals is None else _f32(normals, (self.N, 3))
        st = None if strata is None else _f32(strata)
        _chk(self.L.slamgpu_shard_update(self.h, _ptr(zf), _ptr(idf), zf.shape[0], _ptr(zn), zn.shape[0], _ptr(R), _ptr(nm), _ptr(st)))

    def shard_set_totals_buffer(self, ptr):
        _chk(self.L.slamgpu_shard_set_totals_buffer(self.h, ptr))

    def shard_block_totals(self):
        """device pointer of this shard's [w(nb) | w2(nb)] block totals, nb"""
        t, nb = C.c_void_p(), C.c_int32()
        _chk(self.L.slamgpu_shard_block_totals(self.h, C.byref(t), C.byref(nb)))
        return t.value, nb.value

    def shard_plan(self, gtot_ptr, nb_global, n_shards):
        plan = ShardPlan()
        _chk(self.L.slamgpu_shard_plan(self.h, gtot_ptr, nb_global, n_shards, C.byref(plan)))
        return plan

    def record_floats(self):
        return self.L.slamgpu_shard_record_floats(self.h)

    def shard_pack(self, gtot_ptr, nb_global, n_shards, shard, plan, send_ptr):
        sc = np.zeros(n_shards, np.int64)
        rc = np.zeros(n_shards, np.int64)
        _chk(self.L.slamgpu_shard_pack(self.h, gtot_ptr, nb_global, n_shards, shard, C.byref(plan), send_ptr, _ptr(sc), _ptr(rc)))
        return sc, rc

    def shard_unpack(self, recv_ptr, n_shards, shard, plan):
        _chk(self.L.slamgpu_shard_unpack(self.h, recv_ptr, n_shards, shard, C.byref(plan)))

    def shard_finish(self, plan):
        _chk(self.L.slamgpu_shard_finish(self.h, C.byref(plan)))

    def shard_estimate(self):
        e = np.zeros(4, np.float64)
        _chk(self.L.slamgpu_shard_estimate(self.h, _ptr(e)))
        return e

    def dev_alloc(self, nbytes):
        p = C.c_void_p()
        _chk(self.L.slamgpu_dev_alloc(self.h, nbytes, C.byref(p)))
        return p.value

    def dev_free(self, ptr):
        _chk(self.L.slamgpu_dev_free(self.h, ptr))

    def dev_copy(self, dst, src, nbytes, wait=True):
        if wait:
            _chk(self.L.slamgpu_dev_copy(self.h, dst, src, nbytes))
        else:
            _chk(self.L.slamgpu_dev_copy_async(self.h, dst, src, nbytes))

    # ---- distributed operation (include/slamgpu.h: slamgpu_dist_*) ----
    def dist_export(self):
        blob = C.create_string_buffer(self.L.slamgpu_dist_export_size())
        _chk(self.L.slamgpu_dist_export(self.h, blob))
        return blob.raw

    def dist_connect(self, n_shards, shard, blobs):
        raw = b"".join(blobs)
        assert len(raw) == n_shards * self.L.slamgpu_dist_export_size()
        _chk(self.L.slamgpu_dist_connect(self.h, n_shards, shard, raw))

    def prepare_dist_step(self, controls, Q, dt, zf, idf, zn, R, record_estimate=True):
        ctl = _f32(controls).reshape(-1, 3)
        Q = _f32(Q, 4)
        zf = _f32(zf).reshape(-1, 2)
        zn = _f32(zn).reshape(-1, 2)
        idf = np.ascontiguousarray(idf, np.int32)
        R = _f32(R, 4)
        keep = (ctl, Q, zf, zn, idf, R)
        args = (self.h, _ptr(ctl), ctl.shape[0], _ptr(Q), C.c_float(dt), _ptr(zf), _ptr(idf), zf.shape[0], _ptr(zn), zn.shape[0],
                _ptr(R), 1 if record_estimate else 0)
        fn = self.L.slamgpu_dist_step

        def call(_keep=keep):
            _chk(fn(*args))
        return call

    def dist_step(self, controls, Q, dt, zf, idf, zn, R, record_estimate=True):
        self.prepare_dist_step(controls, Q, dt, zf, idf, zn, R, record_estimate)()

    def dist_totals(self):
        """(this shard's totals, the gathered table, floats per shard) for the all-gather after the last dist step"""
        a, b, n = C.c_void_p(), C.c_void_p(), C.c_int32()
        _chk(self.L.slamgpu_dist_totals(self.h, C.byref(a), C.byref(b), C.byref(n)))
        return a.value, b.value, n.value

    def dist_comm_init(self, comm_id, n_ranks, rank):
        _chk(self.L.slamgpu_dist_comm_init(self.h, comm_id, n_ranks, rank))

    def dist_set_collective(self, mode):
        """0 / False: all-gather, 1 / True: pushed totals + flag kernel, 2: pushed totals + barrier folded into the next launch"""
        _chk(self.L.slamgpu_dist_set_collective(self.h, int(mode)))

    def dist_handshake_test(self, iters=50):
        """(microseconds per barrier, every peer arrived) -- collective"""
        us, ok = C.c_double(), C.c_int32()
        _chk(self.L.slamgpu_dist_handshake_test(self.h, iters, C.byref(us), C.byref(ok)))
        return us.value, bool(ok.value)

    def dist_handshake_enqueue(self, iters=50):
        """the same barriers, enqueued only (contexts driven by one thread: queue everybody's before anybody waits)"""
        _chk(self.L.slamgpu_dist_handshake_test(self.h, iters, None, None))

    def dist_collective_ok(self):
        ok = C.c_int32()
        _chk(self.L.slamgpu_dist_collective_status(self.h, C.byref(ok)))
        return bool(ok.value)

    def dist_gather(self):
        _chk(self.L.slamgpu_dist_gather(self.h))

    def dist_comm_info(self):
        """(ranks, this rank) of the RCCL communicator inside the library, as RCCL itself reports them"""
        n, r = C.c_int32(), C.c_int32()
        _chk(self.L.slamgpu_dist_comm_info(self.h, C.byref(n), C.byref(r)))
        return n.value, r.value

    def dist_remote_reads(self):
        """particles so far whose ancestor at a resample lived on another shard (read in place over xGMI)"""
        v = C.c_uint64()
        _chk(self.L.slamgpu_dist_remote_reads(self.h, C.byref(v)))
        return v.value

    def dist_settle(self):
        _chk(self.L.slamgpu_dist_settle(self.h))

    def shard_estimate_fetch_full(self, max_count=4096):
        raw = np.zeros((max_count, 4), np.float64)
        neff = np.zeros(max_count, np.float32)
        res = np.zeros(max_count, np.int32)
        st = np.zeros(max_count, np.int32)
        n = C.c_int32()
        _chk(self.L.slamgpu_dist_history_fetch(self.h, _ptr(raw), _ptr(neff), _ptr(res), _ptr(st), max_count, C.byref(n)))
        return raw[:n.value].copy(), neff[:n.value].copy(), res[:n.value].copy(), st[:n.value].copy()

    def shard_estimate_async(self):
        _chk(self.L.slamgpu_shard_estimate_async(self.h))

    def shard_estimate_fetch(self, max_count=4096):
        out = np.zeros((max_count, 4), np.float64)
        n = C.c_int32()
        _chk(self.L.slamgpu_shard_estimate_fetch(self.h, _ptr(out), max_count, C.byref(n)))
        return out[:n.value].copy()

    def set_map(self, lm):
        """landmark map for the device observation front end: lm [2, nlm] (xs, ys)"""
        lm = _f32(lm)
        self._nlm = lm.shape[1]
        _chk(self.L.slamgpu_set_map(self.h, _ptr(lm), self._nlm))

    def observe(self, xtrue, max_range, R, noise=0, r1=None, r2=None):
        """slamgpu_observe: dict(z, vis, zf, idf, zn)"""
        nl = self._nlm
        z, vis = np.zeros((nl, 2), np.float32), np.zeros(nl, np.int32)
        zf, idf, zn = np.zeros((nl, 2), np.float32), np.zeros(nl, np.int32), np.zeros((nl, 2), np.float32)
        nz, m, n = C.c_int32(), C.c_int32(), C.c_int32()
        a1 = None if r1 is None else _f32(np.pad(np.asarray(r1, np.float32), (0, max(0, nl - len(r1)))))
        a2 = None if r2 is None else _f32(np.pad(np.asarray(r2, np.float32), (0, max(0, nl - len(r2)))))
        _chk(self.L.slamgpu_observe(self.h, _ptr(_f32(xtrue, 3)), max_range, _ptr(_f32(R, 4)), noise, _ptr(a1), _ptr(a2), _ptr(z), _ptr(vis),
                                    C.byref(nz), _ptr(zf), _ptr(idf), C.byref(m), _ptr(zn), C.byref(n)))
        return dict(z=z[:nz.value].copy(), vis=vis[:nz.value].copy(), zf=zf[:m.value].copy(), idf=idf[:m.value].copy(), zn=zn[:n.value].copy())

    def prepare_step_observe(self, controls, Q, dt, xtrue, max_range, R, noise=2, r1=None, r2=None, normals=None, strata=None,
                             record_estimate=True):
        """slamgpu_step_observe marshalled once (see prepare_step): k predicts + observation made on the device + update"""
        ctl = _f32(controls).reshape(-1, 3)
        Q = _f32(Q, 4)
        R = _f32(R, 4)
        xt = _f32(xtrue, 3)
        nl = self._nlm
        a1 = None if r1 is None else _f32(np.pad(np.asarray(r1, np.float32), (0, max(0, nl - len(r1)))))
        a2 = None if r2 is None else _f32(np.pad(np.asarray(r2, np.float32), (0, max(0, nl - len(r2)))))
        nm = None if normals is None else _f32(normals, (self.N, 3))
        st = None if strata is None else _f32(strata)
        keep = (ctl, Q, R, xt, a1, a2, nm, st)
        args = (self.h, _ptr(ctl), ctl.shape[0], _ptr(Q), C.c_float(dt), _ptr(xt), C.c_float(max_range), _ptr(R), int(noise), _ptr(a1), _ptr(a2),
                _ptr(nm), _ptr(st), 1 if record_estimate else 0)
        fn = self.L.slamgpu_step_observe

        def call(_keep=keep):
            _chk(fn(*args))
        return call

    def step_observe(self, controls, Q, dt, xtrue, max_range, R, **kw):
        self.prepare_step_observe(controls, Q, dt, xtrue, max_range, R, **kw)()

    def prepare_run_observe(self, controls_per_step, Q, dt, xtrue_per_step, max_range, R, noise=2):
        """slamgpu_run_observe marshalled once: returns a closure that makes the C call (a C++ host holds plain arrays already)"""
        K = len(controls_per_step)
        counts = np.ascontiguousarray([np.asarray(c, np.float32).reshape(-1, 3).shape[0] for c in controls_per_step], np.int32)
        rows = [np.asarray(c, np.float32).reshape(-1, 3) for c in controls_per_step if np.asarray(c).size]
        ctl = _f32(np.concatenate(rows) if rows else np.zeros((0, 3), np.float32))
        xt = _f32(np.asarray(xtrue_per_step, np.float32).reshape(K, 3))
        Q = _f32(Q, 4)
        R = _f32(R, 4)
        keep = (counts, ctl, xt, Q, R)
        args = (self.h, K, _ptr(counts), _ptr(ctl), _ptr(Q), C.c_float(dt), _ptr(xt), C.c_float(max_range), _ptr(R), int(noise))
        fn = self.L.slamgpu_run_observe

        def call(_keep=keep):
            _chk(fn(*args))
        return call

    def run_observe(self, controls_per_step, Q, dt, xtrue_per_step, max_range, R, noise=2):
        """slamgpu_run_observe: K iterations (k predicts + device-made observation + update + estimate each) in one C call;
        controls_per_step: K arrays of (V, G, phi_true) rows; xtrue_per_step: K poses"""
        K = len(controls_per_step)
        counts = np.ascontiguousarray([np.asarray(c, np.float32).reshape(-1, 3).shape[0] for c in controls_per_step], np.int32)
        rows = [np.asarray(c, np.float32).reshape(-1, 3) for c in controls_per_step if np.asarray(c).size]
        ctl = _f32(np.concatenate(rows) if rows else np.zeros((0, 3), np.float32))
        xt = _f32(np.asarray(xtrue_per_step, np.float32).reshape(K, 3))
        _chk(self.L.slamgpu_run_observe(self.h, K, _ptr(counts), _ptr(ctl), _ptr(_f32(Q, 4)), C.c_float(dt), _ptr(xt), C.c_float(max_range),
                                        _ptr(_f32(R, 4)), int(noise)))

    def observe_fetch(self):
        """the observation packet of the last step_observe: dict(z, vis, zf, idf, zn)"""
        nl = self._nlm
        z, vis = np.zeros((nl, 2), np.float32), np.zeros(nl, np.int32)
        zf, idf, zn = np.zeros((nl, 2), np.float32), np.zeros(nl, np.int32), np.zeros((nl, 2), np.float32)
        nz, m, n = C.c_int32(), C.c_int32(), C.c_int32()
        _chk(self.L.slamgpu_observe_fetch(self.h, _ptr(z), _ptr(vis), C.byref(nz), _ptr(zf), _ptr(idf), C.byref(m), _ptr(zn), C.byref(n)))
        return dict(z=z[:nz.value].copy(), vis=vis[:nz.value].copy(), zf=zf[:m.value].copy(), idf=idf[:m.value].copy(), zn=zn[:n.value].copy())

    def associate(self, z, R, gate_reject=4.0, gate_augment=25.0, want_labels=True, mode=ASSOC_AUTO, want_stats=False):
        """per-particle gated nearest-neighbour association (slamgpu_associate_ex): labels [N, nz], consensus [nz], support [nz]
        (+ stats dict: triples evaluated, grid entries, device ms, grid used)"""
        z = _f32(z).reshape(-1, 2)
        nz = z.shape[0]
        lab = np.zeros((self.N, nz), np.int32) if want_labels else None
        cons = np.zeros(nz, np.int32)
        sup = np.zeros(nz, np.float32)
        st = np.zeros(4, np.float64)
        _chk(self.L.slamgpu_associate_ex(self.h, _ptr(z), nz, _ptr(_f32(R, 4)), gate_reject, gate_augment, int(mode), _ptr(lab), _ptr(cons), _ptr(sup),
                                         _ptr(st)))
        if want_stats:
            return lab, cons, sup, dict(triples=st[0], grid_entries=st[1], ms=st[2], grid=bool(st[3]))
        return lab, cons, sup

    _REPORT = ("rewritten", "opened", "reused", "dropped", "slots", "dead", "need", "census")

    def _particle_opt(self, gate_reject, gate_augment, mode, new_share, p_new, census_every, excl=(0.0, 0.0, 2.0)):
        o = ParticleAssoc()
        o.excl_base, o.excl_per_m, o.unique_ratio = (float(v) for v in excl)
        o.gate_reject, o.gate_augment, o.mode = gate_reject, gate_augment, int(mode)
        o.new_share, o.p_new, o.census_every = new_share, p_new, int(census_every)
        return o

    def update_particle(self, z, R, gate_reject=4.0, gate_augment=25.0, mode=ASSOC_AUTO, new_share=0.0, p_new=1.0, census_every=1,
                        normals=None, strata=None, excl=(0.0, 0.0, 2.0)):
        """one observation step in which every particle acts on its own gated association (slamgpu_update_particle); returns the report"""
        z = _f32(z).reshape(-1, 2)
        o = self._particle_opt(gate_reject, gate_augment, mode, new_share, p_new, census_every, excl)
        nm = None if normals is None else _f32(normals, (self.N, 3))
        st = None if strata is None else _f32(strata)
        rep = np.zeros(8, np.int32)
        _chk(self.L.slamgpu_update_particle(self.h, _ptr(z), z.shape[0], _ptr(_f32(R, 4)), C.byref(o), _ptr(nm), _ptr(st), _ptr(rep)))
        return dict(zip(self._REPORT, (int(v) for v in rep)))

    def update_labels(self, z, R, labels, new_share=0.0, p_new=1.0, census_every=1, normals=None, strata=None):
        """the same step with the caller's labels [N, nz] (slamgpu_update_labels)"""
        z = _f32(z).reshape(-1, 2)
        lab = np.ascontiguousarray(labels, np.int32).reshape(self.N, z.shape[0])
        o = self._particle_opt(0.0, 0.0, ASSOC_AUTO, new_share, p_new, census_every)
        nm = None if normals is None else _f32(normals, (self.N, 3))
        st = None if strata is None else _f32(strata)
        rep = np.zeros(8, np.int32)
        _chk(self.L.slamgpu_update_labels(self.h, _ptr(z), z.shape[0], _ptr(_f32(R, 4)), _ptr(lab), C.byref(o), _ptr(nm), _ptr(st), _ptr(rep)))
        return dict(zip(self._REPORT, (int(v) for v in rep)))

    def debug_stamps(self, max_blocks=8192):
        out = np.zeros((max_blocks, 16), np.uint64)
        n = C.c_int32()
        _chk(self.L.slamgpu_debug_stamps(self.h, _ptr(out), max_blocks, C.byref(n)))
        return out[:n.value].copy()

    def stream(self):
        return self.L.slamgpu_stream(self.h)

    def timer_start(self):
        _chk(self.L.slamgpu_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_double()
        _chk(self.L.slamgpu_timer_stop(self.h, C.byref(ms)))
        return ms.value

    def profile(self, enable=True):
        _chk(self.L.slamgpu_profile(self.h, int(enable)))

    def kernel_time(self, name):
        ms, n = C.c_double(), C.c_int64()
        _chk(self.L.slamgpu_kernel_time(self.h, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value
