import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(ROOT, "data")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _hip_runtime_up(request):
    """The HIP runtime re-seeds libc's rand() when it initialises (measured: tools/rand_probe.py: the first context of a process,
    nothing afterwards).  The oracle and the host front end draw the reference's random numbers from libc rand(), as the
    reference does, so a GPU test that seeds an oracle run and only THEN creates its first context would be driven by a tape that
    differs from run to run.  GPU sessions bring the runtime up first, once.  (The product does the same where it matters:
    slam-backend seeds after the context exists, slam_backend.cpp.)"""
    if not any(item.get_closest_marker("gpu") for item in request.session.items):
        return
    import slam_amd
    if slam_amd.device_count() >= 1:
        slam_amd.SlamGpu(256, 4, method=1, n_effective=192, rng_mode=slam_amd.RNG_PHILOX, seed=1).close()


@pytest.fixture(scope="session")
def oracle():
    from oracle import orc
    orc.build_oracle()
    return orc.Oracle()


@pytest.fixture(scope="session")
def kat():
    return np.load(os.path.join(GOLDEN, "kat_functions.npz"))


def load_golden(name):
    # materialised: an NpzFile re-reads and decompresses the member on every [] access
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def sim_args(mapname, method, N, seed, extra=()):
    return ["-m", os.path.join(DATA, mapname + ".mat"), "-method", method, "-NPARTICLES", N,
            "-NEFFECTIVE", int(0.75 * N), "-SWITCH_SEED_RANDOM", seed] + list(extra)


def bits_equal(a, b):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))
