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
    # The simulator runs the kernel for an under-filled GPU (eleven hundred fibers that poll
    # each other) several times slower than the one for a full GPU: simulator tests use the
    # latter unless they ask for both (both_glz_kernels below).
    g.L.spice_gpu_glz_lone_max_jobs(0)
    return g


@pytest.fixture(scope="session")
def gpu():
    """The real library on the real device; GPU tests fail loudly without it."""
    import spice_amd
    return spice_amd.load()


@pytest.fixture(params=["kernel for a full GPU", "kernel for an under-filled GPU"])
def both_glz_kernels(request):
    """GLZ parity tests that take this fixture run twice: every launch through the kernel that
    fits eight images on a CU, then every launch through the one that gives an image a CU of its
    own (glz_kernels.hip / glz_lone.hip).  The streams must not differ.  Applies to whichever
    library the test uses (simulator build and real one are set alike)."""
    import spice_amd
    from spice_amd.binding import SpiceGpu
    n = 0 if request.param == "kernel for a full GPU" else 1 << 30
    libs = []
    so = os.path.join(SIM_DIR, "libspicegpu_sim.so")
    if os.path.exists(so):
        libs.append((ctypes.CDLL(so), 0))                      # back to the simulator tests' default
    try:
        libs.append((ctypes.CDLL(spice_amd.LIB_PATH), -1))     # back to the library's own default
    except OSError:
        pass
    for lib, _ in libs:
        lib.spice_gpu_glz_lone_max_jobs.argtypes = [ctypes.c_int]
        lib.spice_gpu_glz_lone_max_jobs(n)
    yield request.param
    for lib, default in libs:
        lib.spice_gpu_glz_lone_max_jobs(default)
