"""The shipped library loads without a GPU and exports every function the
boundary header declares; no compute entry point is called here."""
import ctypes
import os
import re

import spice_amd
from spice_amd import binding as B

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "spice_gpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    for _ in range(3):                       # drop struct / enum bodies
        src = re.sub(r"\{[^{}]*\}", "{}", src)
    return re.findall(r"^[A-Za-z_][A-Za-z0-9_ \*]*?\b([a-z_][A-Za-z0-9_]*)\s*\(", src, flags=re.M)


def test_library_exports_every_declared_function():
    assert os.path.exists(spice_amd.LIB_PATH), "build spice_amd/csrc first"
    lib = ctypes.CDLL(spice_amd.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 50 and "glz_encode" in names and "mjpeg_encoder_end_frame" in names
    assert [n for n in names if not hasattr(lib, n)] == []


def test_binding_list_matches_header():
    assert sorted(B.EXPORTS) == sorted(declared_functions())
    B.SpiceGpu(ctypes.CDLL(spice_amd.LIB_PATH))      # attaches every prototype; no init, no GPU


def test_state_struct_sizes_agree():
    from oracle import oracle as O
    assert ctypes.sizeof(B.SpiceGpuMjpegRateState) == ctypes.sizeof(O.MjpegRateState) == 232
