"""The documented reference-side bindings, compiled and run (round-3 review item 8).

tests/cabi/accel_shim.h            the AcceleratorHandler a maintainer drops in for the FPGA one (AcceleratorHandler.h:10-23)
tests/cabi/fastslam2gpu_adapter.h  the device-backed algorithm object with FastSLAM2's methods (fastslam2.h:20-31)
tests/cabi/cabi_driver.cpp         plays the reference's callers: packs the accelerator window exactly as computeJacobians
                                   does (core.cpp:586-664), drives the adapter as FastSLAM2Wrapper::run does
                                   (fastslam2wrapper.cpp:51-117)
g++ only, against include/slamgpu.h; checked against the reference-held vectors (jac_* KATs, the first golden steps)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import load_golden, sim_args

pytestmark = pytest.mark.gpu
f32 = np.float32
HERE = os.path.dirname(os.path.abspath(__file__))
RM = np.array([[0.1 ** 2, 0], [0, 0.017453292519943 ** 2]], f32)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


@pytest.fixture(scope="module")
def drv():
    import slam_amd
    assert slam_amd.device_count() >= 1, "GPU tests need a HIP device"
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "cabi")])
    L = C.CDLL(os.path.join(HERE, "cabi", "libcabi_driver.so"))
    L.cabi_last_error.restype = C.c_char_p
    L.cabi_algo_create.restype = C.c_void_p
    L.cabi_algo_create.argtypes = [C.c_int] * 6 + [C.c_float, C.c_float, C.c_int, C.c_int, C.c_uint]
    L.cabi_algo_destroy.argtypes = [C.c_void_p]
    L.cabi_algo_predict.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_float]
    L.cabi_algo_update.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    L.cabi_algo_estimate.argtypes = [C.c_void_p, C.c_void_p]
    L.cabi_algo_landmarks.argtypes = [C.c_void_p]
    L.cabi_algo_fetch.argtypes = [C.c_void_p] * 4
    L.cabi_compute_jacobians.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 4
    return L


def test_accelerator_shim_runs_the_reference_window(drv, kat):
    """computeJacobians through the AcceleratorHandler drop-in: window packed and unpacked by the caller's code path
    (Eigen linear order in, 16 floats per feature out), one landmark at a time and batched with an idf selection."""
    n = kat["jac_xv"].shape[0]
    for i in range(n):
        zp, Hv, Hf, Sf = np.zeros(2, f32), np.zeros(6, f32), np.zeros(4, f32), np.zeros(4, f32)
        idf = np.array([0], np.int32)
        rc = drv.cabi_compute_jacobians(_p(kat["jac_xv"][i]), _p(RM), _p(np.ascontiguousarray(kat["jac_xf"][i])), _p(np.ascontiguousarray(kat["jac_Pf"][i])), 1,
                                        _p(idf), 1, _p(zp), _p(Hv), _p(Hf), _p(Sf))
        assert rc == 0, drv.cabi_last_error()
        np.testing.assert_allclose(zp, kat["jac_zp"][i], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(Hv.reshape(2, 3), kat["jac_Hv"][i], rtol=2e-6, atol=1e-9)
        np.testing.assert_allclose(Hf.reshape(2, 2), kat["jac_Hf"][i], rtol=2e-6, atol=1e-9)
        np.testing.assert_allclose(Sf.reshape(2, 2), kat["jac_Sf"][i], rtol=1e-5, atol=1e-9)
    # batched, with a selection that permutes and skips (idf picks landmarks out of the particle's map, core.cpp:608-617):
    # row k of the output must be the single-landmark result for landmark idf[k] at the same pose
    xv = kat["jac_xv"][0]
    idf = np.array([5, 0, 17, 3, 95, 40], np.int32)
    k = idf.size
    zp, Hv, Hf, Sf = np.zeros((k, 2), f32), np.zeros((k, 6), f32), np.zeros((k, 4), f32), np.zeros((k, 4), f32)
    rc = drv.cabi_compute_jacobians(_p(xv), _p(RM), _p(np.ascontiguousarray(kat["jac_xf"])), _p(np.ascontiguousarray(kat["jac_Pf"])), n, _p(idf), k,
                                    _p(zp), _p(Hv), _p(Hf), _p(Sf))
    assert rc == 0, drv.cabi_last_error()
    for r, j in enumerate(idf):
        one = [np.zeros(2, f32), np.zeros(6, f32), np.zeros(4, f32), np.zeros(4, f32)]
        rc = drv.cabi_compute_jacobians(_p(xv), _p(RM), _p(np.ascontiguousarray(kat["jac_xf"][j])), _p(np.ascontiguousarray(kat["jac_Pf"][j])), 1,
                                        _p(np.array([0], np.int32)), 1, *[_p(a) for a in one])
        assert rc == 0
        assert np.array_equal(zp[r], one[0]) and np.array_equal(Hv[r], one[1]) and np.array_equal(Hf[r], one[2]) and np.array_equal(Sf[r], one[3])


@pytest.mark.parametrize("math_mode", [0, 1], ids=["strict", "fast"])
@pytest.mark.parametrize("name,mapname,method,N,seed", [("traj_fs2_webmap_N100_s7", "example_webmap", 2, 100, 7),
                                                          ("traj_fs2_loop902_N100_s3", "example_loop902", 2, 100, 3),
                                                          ("traj_fs1_webmap_N100_s7", "example_webmap", 1, 100, 7)])
def test_adapter_follows_the_golden_steps(drv, name, mapname, method, N, seed, math_mode):
    """The algorithm-object adapter driven like the wrapper's loop, libc rand() tape in the reference's order: the first
    golden observation steps (before any ancestor can differ: particle states against the reference's snapshots 1..3,
    estimates until the first resample)."""
    from slam_amd import host
    g = load_golden(name)
    h = drv.cabi_algo_create(method, N, 120, int(g["meta_n_effective"]), int(g["meta_use_heading"]), int(g["meta_add_predict_noise"]),
                             float(g["meta_wheel_base"]), float(g["meta_sigma_phi"]), 1, math_mode, 0)
    assert h, drv.cabi_last_error()
    sim = host.HostSim(sim_args(mapname, "FASTSLAM2" if method == 2 else "FASTSLAM1", N, seed))   # seeds rand() AFTER the context exists
    Q, R, dt = sim.noise()
    first_res = int(np.argmax(g["resampled"])) if g["resampled"].any() else 10 ** 9
    k, nf = 0, 0
    while k < min(12, first_res + 1):
        r, V, G, phi = sim.control()
        assert r >= 0
        x = sim.true_pose()
        assert drv.cabi_algo_predict(h, _p(x), V, G, _p(Q), float(dt)) == 0, drv.cabi_last_error()
        if r != 1:
            continue
        nf_dev = drv.cabi_algo_landmarks(h)
        assert nf_dev == nf
        zf, idf, zn = sim.observe(nf)
        assert drv.cabi_algo_update(h, _p(zf), _p(idf), zf.shape[0], _p(zn), zn.shape[0], _p(R)) == 0, drv.cabi_last_error()
        nf += zn.shape[0]
        k += 1
        e = np.zeros(3)
        assert drv.cabi_algo_estimate(h, _p(e)) == 0, drv.cabi_last_error()
        # (k = first_res + 1 is the first step that resamples: a stratum on the other side of a cumulative-sum boundary picks a
        # neighbouring ancestor there, and the mean moves by that particle's share)
        assert np.abs(e[:2] - g["est"][k - 1, :2]).max() <= (1e-3 if k <= first_res else 1e-2), (k, e, g["est"][k - 1])
        if k in (1, 2, 3) and not g["resampled"][k - 1]:
            xv, w, xf = np.zeros((N, 3), f32), np.zeros(N, f32), np.zeros((N, nf, 2), f32)
            assert drv.cabi_algo_fetch(h, _p(xv), _p(w), _p(xf)) == 0, drv.cabi_last_error()
            exp = {key: g["snap%d_post_%s" % (k, key)] for key in ("xv", "w", "xf")}
            assert np.abs(xv - exp["xv"]).max() <= 2e-4, (k, np.abs(xv - exp["xv"]).max())
            assert np.abs(xf - exp["xf"]).max() <= 1e-3, k
            rel = np.abs(w.astype(np.float64) / w.sum(dtype=np.float64) / (exp["w"].astype(np.float64) / exp["w"].sum(dtype=np.float64)) - 1)
            assert np.median(rel) <= 1e-2 and rel.max() <= 0.25, (k, np.median(rel), rel.max())
    assert k >= 3
    drv.cabi_algo_destroy(h)
    sim.close()


# ---- the reference's OWN code on the accelerator seam ----------------------------------------------------------------------
# oracle/_ref/libslamref_accel.so = the reference's numeric core compiled (authoring container, oracle/Makefile: refaccel) the
# way the reference builds for its FPGA, -DJACOBIAN_ACCELERATOR, with tests/cabi/accel_shim.h as its AcceleratorHandler.  The
# file travels to the GPU box prebuilt; here the reference's computeJacobians (core.cpp:586-664) packs its own window and the
# numbers come from slamgpu_jacobians.
@pytest.fixture(scope="module")
def ref_accel():
    from oracle import orc
    if not os.path.exists(orc.REF_ACCEL_SO):
        pytest.skip("oracle/_ref/libslamref_accel.so not built (authoring container: make -C oracle refaccel)")
    return orc.Reference(accel=True)


def test_reference_compute_jacobians_on_the_gpu(ref_accel, kat):
    n = kat["jac_xv"].shape[0]
    for i in range(n):
        zp, Hv, Hf, Sf = [x[0] for x in ref_accel.compute_jacobians(kat["jac_xv"][i], RM, kat["jac_xf"][i:i + 1], kat["jac_Pf"][i:i + 1])]
        np.testing.assert_allclose(zp, kat["jac_zp"][i], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(Hv, kat["jac_Hv"][i], rtol=2e-6, atol=1e-9)
        np.testing.assert_allclose(Hf, kat["jac_Hf"][i], rtol=2e-6, atol=1e-9)
        np.testing.assert_allclose(Sf, kat["jac_Sf"][i], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("name,mapname,method,N,seed", [("traj_fs2_webmap_N100_s7", "example_webmap", "FASTSLAM2", 100, 7),
                                                          ("traj_fs1_webmap_N100_s7", "example_webmap", "FASTSLAM1", 100, 7)])
def test_reference_filter_with_jacobians_on_the_gpu(ref_accel, name, mapname, method, N, seed):
    """The reference's FastSLAM run -- its own predict / sampleProposal / likelihood / featureUpdate / resample code -- with every
    computeJacobians call (2 m + 1 per particle and step in FastSLAM 2) served by the GPU through the drop-in handler: the use
    the reference's README describes for its FPGA.  Same libc rand() stream, Jacobians within an ulp or two of the CPU
    branch's: the run must follow the golden trajectory of the CPU build until a resample can pick a different ancestor."""
    g = load_golden(name)
    r = ref_accel.sim(sim_args(mapname, method, N, seed))
    first_res = int(np.argmax(g["resampled"])) if g["resampled"].any() else 10 ** 9
    k = 0
    while k < 40:
        a = r.control()
        assert a >= 0
        if a != 1:
            continue
        r.observe()
        k += 1
        est = r.estimate()
        ob = r.last_obs()
        assert ob["zf"].shape[0] == g["m"][k - 1] and ob["zn"].shape[0] == g["n"][k - 1]
        assert np.array_equal(ob["zf"], g["zf"][k - 1, :g["m"][k - 1]])   # the simulator side draws the same rand() values
        if k <= first_res + 1:
            assert np.abs(est[:2] - g["est"][k - 1, :2]).max() <= (1e-3 if k <= first_res else 1e-2), (k, est, g["est"][k - 1])
            if k <= first_res:
                p = r.particles()
                assert np.abs(p["xv"][:8] - g["xv_head"][k - 1]).max() <= 2e-4, k
                rel = np.abs(p["w"][:8].astype(np.float64) / g["w_head"][k - 1] - 1)
                assert rel.max() <= (0.16 if method == "FASTSLAM2" else 1e-3), (k, rel.max())
        else:
            assert np.hypot(*(est[:2] - g["true"][k - 1, :2])) <= np.hypot(*(g["est"][k - 1, :2] - g["true"][k - 1, :2])) + 0.5, k
    r.close()


def test_multiparticle_form_of_the_accelerator_window(kat):
    """The MULTIPARTICLE_ACCELERATOR form of seam 1 (AcceleratorHandler.h:17-21: setParticlesCount + start): the drop-in compiled
    with that macro, its window written and read back as FastSLAM2::precomputeAllLikelihoodGivenXv does (fastslam2.cpp:172-286:
    one self-describing record per particle and re-observed landmark).  Every record must carry exactly what the single-particle
    window gives for that particle and landmark (bit for bit: the same device function), and the reference-held Jacobian vectors
    within the seam's bound; a record that runs beyond the buffer is refused."""
    from slam_amd import capi
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "cabi")])
    L = C.CDLL(os.path.join(HERE, "cabi", "libcabi_multi.so"))
    L.cabi_multi_last_error.restype = C.c_char_p
    L.cabi_multi_window.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 4
    n = kat["jac_xv"].shape[0]
    P, nf = 24, n
    rng = np.random.default_rng(3)
    xv = np.ascontiguousarray(kat["jac_xv"][rng.integers(0, n, P)], f32)
    xv[:, :2] += rng.normal(0, 0.05, (P, 2)).astype(f32)
    xf = np.ascontiguousarray(np.broadcast_to(kat["jac_xf"], (P, nf, 2)), f32)
    Pf = np.ascontiguousarray(np.broadcast_to(kat["jac_Pf"], (P, nf, 2, 2)), f32)
    idf = np.array([5, 0, 17, 3, 95, 40, 41], np.int32)
    k = idf.size
    zp, Hv, Hf, Sf = np.zeros((P, k, 2), f32), np.zeros((P, k, 6), f32), np.zeros((P, k, 4), f32), np.zeros((P, k, 4), f32)
    rc = L.cabi_multi_window(_p(xv), _p(RM), _p(xf), _p(Pf), P, nf, _p(idf), k, _p(zp), _p(Hv), _p(Hf), _p(Sf))
    assert rc == 0, L.cabi_multi_last_error()
    for p in range(P):
        z1, hv1, hf1, sf1 = capi.jacobians(xv[p], RM, xf[p][idf], Pf[p][idf])
        assert np.array_equal(zp[p], z1) and np.array_equal(Hv[p].reshape(k, 2, 3), hv1)
        assert np.array_equal(Hf[p].reshape(k, 2, 2), hf1) and np.array_equal(Sf[p].reshape(k, 2, 2), sf1)
    # the reference-held vectors themselves, one record each (pose i, landmark i)
    xv2 = np.ascontiguousarray(kat["jac_xv"], f32)
    for i in range(0, n, 7):
        out = [np.zeros((1, 1, 2), f32), np.zeros((1, 1, 6), f32), np.zeros((1, 1, 4), f32), np.zeros((1, 1, 4), f32)]
        rc = L.cabi_multi_window(_p(xv2[i:i + 1]), _p(RM), _p(np.ascontiguousarray(kat["jac_xf"][i:i + 1])), _p(np.ascontiguousarray(kat["jac_Pf"][i:i + 1])),
                                 1, 1, _p(np.array([0], np.int32)), 1, *[_p(a) for a in out])
        assert rc == 0, L.cabi_multi_last_error()
        np.testing.assert_allclose(out[0].ravel(), kat["jac_zp"][i], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(out[1].reshape(2, 3), kat["jac_Hv"][i], rtol=2e-6, atol=1e-9)
        np.testing.assert_allclose(out[2].reshape(2, 2), kat["jac_Hf"][i], rtol=2e-6, atol=1e-9)
        np.testing.assert_allclose(out[3].reshape(2, 2), kat["jac_Sf"][i], rtol=1e-5, atol=1e-9)
    # the C entry on its own: a window whose last record claims more features than the buffer holds is refused
    lib = capi.load_library()
    win = np.zeros(30, f32)
    win[0] = 2.0
    assert lib.slamgpu_jacobians_multi(_p(win), 1, win.size) < 0 and b"beyond the window" in lib.slamgpu_last_error()
    win[0] = 1.0
    assert lib.slamgpu_jacobians_multi(_p(win), 1, win.size) == 0
    assert lib.slamgpu_jacobians_multi(_p(win), 0, 0) == 0
