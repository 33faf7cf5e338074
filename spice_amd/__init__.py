"""MI355X-native image compression path of the SPICE display server.

`load()` binds the hand-written HIP library (spice_amd/csrc/libspicegpu.so).
There is no CPU implementation behind it: a missing library or a machine
without a usable GPU raises.
"""
import ctypes
import os

from .binding import SpiceGpu, SpiceGpuError  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libspicegpu.so")
_gpu = None


def load(device=None):
    """Returns the process-wide binding, initialised on `device` (default:
    LOCAL_RANK, else 0)."""
    global _gpu
    if _gpu is None:
        if not os.path.exists(LIB_PATH):
            raise SpiceGpuError(
                "%s is missing: build it with `make -C spice_amd/csrc` (hipcc, gfx950)" % LIB_PATH)
        gpu = SpiceGpu(ctypes.CDLL(LIB_PATH))
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        gpu.init(device)
        _gpu = gpu
    return _gpu
