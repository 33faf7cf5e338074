import ctypes
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SIM_DIR = os.path.join(ROOT, "tests", "sim")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def sim():
    """The product sources compiled for the CPU wave simulator (test build)."""
    from spice_amd.binding import SpiceGpu
    so = os.path.join(SIM_DIR, "libspicegpu_sim.so")
    subprocess.check_call(["make", "-s", "-C", SIM_DIR])
    g = SpiceGpu(ctypes.CDLL(so))
    g.init(0)
    return g


@pytest.fixture(scope="session")
def gpu():
    """The real library on the real device; GPU tests fail loudly without it."""
    import spice_amd
    return spice_amd.load()
