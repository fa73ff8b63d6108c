"""Observation front end on the device: slamgpu_set_map / slamgpu_observe =
getObservations + addObservationNoise + dataAssociationKnown (core.cpp:185-273, 438-449, 91-120) as one kernel, against the
product's host front end (libslamhost), which reproduces the reference's observation tape bit for bit
(tests/test_host_frontend.py::test_observation_tape_matches_reference).

Visible ids, counts, the re-observed / new split and the feature indices must be identical; ranges are bit-identical
(double sqrt rounded once on both sides); bearings agree to 1 ulp (the device rounds a double atan2 once, the reference calls
glibc's atan2f); the sensor noise is applied with the same float operations, given the same normals."""
import os

import numpy as np
import pytest

from conftest import DATA, sim_args

pytestmark = pytest.mark.gpu
f32 = np.float32


@pytest.fixture(scope="module")
def sg():
    import slam_amd
    assert slam_amd.device_count() >= 1
    return slam_amd


@pytest.mark.parametrize("mapname,extra", [("example_webmap", []), ("example_loop1", []), ("synthetic", ["-MAX_RANGE", 25])])
def test_device_front_end_matches_the_host_front_end(sg, tmp_path, mapname, extra):
    from slam_amd import host
    if mapname == "synthetic":
        lm = host.synthetic_landmarks(99, 3000, -130, 100, -100, 90)
        h0 = host.HostSim(sim_args("example_webmap", "FASTSLAM2", 100, 7))
        _, wp = h0.map()
        h0.close()
        mp = str(tmp_path / "syn3000.mat")
        host.write_map(mp, lm, wp)
        open(str(tmp_path / "syn3000.ini"), "w").write(open(os.path.join(DATA, "example_webmap.ini")).read())
        args = ["-m", mp, "-method", "FASTSLAM2", "-SWITCH_SEED_RANDOM", 7, "-SWITCH_SENSOR_NOISE", 0] + extra
    else:
        args = sim_args(mapname, "FASTSLAM2", 100, 7) + ["-SWITCH_SENSOR_NOISE", 0] + extra
    h = host.HostSim(args)
    lm, _ = h.map()
    Q, R, dt = h.noise()
    s = sg.SlamGpu(256, max(h.nlm, 1), method=2, rng_mode=sg.RNG_PHILOX)
    s.set_map(lm)
    nf, k, worst_ulp, seen_new, seen_old = 0, 0, 0.0, 0, 0
    while k < 150:
        r, V, G, phi = h.control()
        if r < 0:
            break
        if r == 1:
            zf, idf, zn = h.observe(nf)
            z, vis = h.last_z()
            got = s.observe(h.true_pose(), float(h.conf.MAX_RANGE), R, noise=0)
            assert np.array_equal(got["vis"], vis), k
            assert np.array_equal(got["z"][:, 0].view(np.uint32), z[:, 0].view(np.uint32)), k      # ranges: same bits
            # bearing = atan2 - phi: one ulp of the atan2 (|atan2| <= pi: 2.4e-7) survives the subtraction unchanged
            dif = np.abs(got["z"][:, 1].astype(np.float64) - z[:, 1]) / 2.384185791015625e-07
            worst_ulp = max(worst_ulp, float(dif.max()) if dif.size else 0.0)
            assert np.array_equal(got["idf"], idf), k
            assert got["zf"].shape == zf.shape and got["zn"].shape == zn.shape, k
            np.testing.assert_allclose(got["zf"], zf, rtol=0, atol=1e-6)
            np.testing.assert_allclose(got["zn"], zn, rtol=0, atol=1e-6)
            nf += zn.shape[0]
            seen_new += zn.shape[0]
            seen_old += zf.shape[0]
            k += 1
    assert worst_ulp <= 1.0 + 1e-9, worst_ulp
    assert seen_new >= 3 and seen_old >= 50
    s.close()
    h.close()


def test_device_front_end_sensor_noise(sg):
    """tape noise: z(0,c) += r1[c] * sqrt(R00), z(1,c) += r2[c] * sqrt(R11) in visibility order (core.cpp:438-449), float32;
    Philox noise: different per landmark and per step, zero-mean at the configured scale."""
    from slam_amd import host
    h = host.HostSim(sim_args("example_webmap", "FASTSLAM2", 100, 7) + ["-SWITCH_SENSOR_NOISE", 0])
    lm, _ = h.map()
    _, R, _ = h.noise()
    s = sg.SlamGpu(256, h.nlm, method=2, rng_mode=sg.RNG_PHILOX, seed=11)
    pose = np.array([10.0, -5.0, 0.3], f32)
    s.set_map(lm)
    clean = s.observe(pose, 60.0, R, noise=0)
    nz = clean["z"].shape[0]
    assert nz >= 3
    rng = np.random.default_rng(3)
    r1, r2 = rng.normal(size=nz).astype(f32), rng.normal(size=nz).astype(f32)
    s.set_map(lm)  # fresh association table
    noisy = s.observe(pose, 60.0, R, noise=1, r1=r1, r2=r2)
    exp0 = clean["z"][:, 0] + r1 * np.sqrt(R[0, 0], dtype=f32)
    exp1 = clean["z"][:, 1] + r2 * np.sqrt(R[1, 1], dtype=f32)
    assert np.array_equal(noisy["z"][:, 0].view(np.uint32), exp0.astype(f32).view(np.uint32))
    assert np.array_equal(noisy["z"][:, 1].view(np.uint32), exp1.astype(f32).view(np.uint32))
    s.set_map(lm)
    d = []
    for _ in range(200):
        p = s.observe(pose, 60.0, R, noise=2)
        d.append(p["z"] - clean["z"])
    d = np.stack(d)
    assert abs(d[..., 0].mean()) < 0.02 and 0.08 < d[..., 0].std() < 0.12      # sigmaR = 0.1 m
    assert 0.8 < d[..., 1].std() / 0.017453292519943 < 1.2                      # sigmaB = 1 degree
    s.close()
    h.close()
