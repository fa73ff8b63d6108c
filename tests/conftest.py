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
