"""The C host side (spice_amd/host/red_compress_host.c: red_worker.c's call
sequences in C, linked against the library like spice-server would be): a
stream of bitmaps through host_compress_image, compared with the composition
of the oracle's parts.  The CPU test links the wave-simulator build of the
library; the GPU test links libspicegpu.so itself."""
import os
import struct
import subprocess

import numpy as np
import pytest

from spice_amd import binding as B

from helpers import assert_same_stream, split_rows
from test_compress_image import bitmaps, expected

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "spice_amd", "host")
TYPE_NAMES = {1: "QUIC", 101: "LZ_RGB", 102: "GLZ_RGB", 105: "JPEG", 107: "ZLIB_GLZ_RGB", 108: "JPEG_ALPHA"}


def run_cli(exe, tmp_path, oracle, comp_mode, zlib_wrap, n):
    cases = []
    script = [b"HOST1", struct.pack("<iIiiiiI", comp_mode, 60000, 1, 85, int(zlib_wrap), 3, n)]
    for k, (fmt, w, h, px, unstable, can_lossy) in enumerate(bitmaps(n)):
        chunks = split_rows(px, 1 + k % 3)
        top_down = k % 4 != 3
        cases.append((k, fmt, w, h, px, chunks, top_down, unstable, can_lossy))
        script.append(struct.pack("<iiiiIIII", fmt, int(top_down), int(unstable), int(can_lossy), w, h,
                                  px.shape[1], len(chunks)))
        script.append(struct.pack("<%dI" % len(chunks), *[c.shape[0] for c in chunks]))
        script.extend(c.tobytes() for c in chunks)
    src, dst = str(tmp_path / "script.bin"), str(tmp_path / "result.bin")
    open(src, "wb").write(b"".join(script))
    subprocess.run([exe, src, dst], check=True, timeout=600)
    blob = open(dst, "rb").read()
    od = oracle.GlzDict(60000, 1)
    seen, pos = {}, 0
    try:
        for (k, fmt, w, h, px, chunks, top_down, unstable, can_lossy) in cases:
            ok, typ, size, glz_size, jpeg_size, nfreed = struct.unpack_from("<iiIIII", blob, pos)
            pos += 24
            freed = list(struct.unpack_from("<%dQ" % nfreed, blob, pos))
            pos += 8 * nfreed
            data = np.frombuffer(blob, dtype=np.uint8, count=size, offset=pos)
            pos += size
            # the C side tags images 1000 + k; the oracle's window is fed 100 + k
            want = expected(oracle, od, comp_mode, zlib_wrap, k, fmt, w, h, px, chunks, top_down,
                            unstable, can_lossy)
            assert [t - 900 for t in freed] == od.pop_freed(), k
            what = "bitmap %d fmt %d %dx%d -> %s" % (k, fmt, w, h, want and want[0])
            if want is None:
                assert ok == 0, what
                seen["raw"] = seen.get("raw", 0) + 1
                continue
            assert TYPE_NAMES[typ] == want[0], what
            seen[want[0]] = seen.get(want[0], 0) + 1
            if want[1] is None:
                # QUIC, LZ, JPEG_ALPHA: HOST_COMPRESS_DELEGATE, never "compressed"
                assert ok == 2 and size == 0 and jpeg_size == 0, what
            else:
                assert ok == 1, what
                assert_same_stream(want[1], data, what)
                if want[0] == "ZLIB_GLZ_RGB":
                    assert glz_size == want[2], what
        assert pos == len(blob)
    finally:
        od.close()
    return seen


def test_sim_c_host_stream(tmp_path, sim, oracle):
    subprocess.check_call(["make", "-s", "-C", HOST, "sim"])
    seen = run_cli(os.path.join(HOST, "host_cli_sim"), tmp_path, oracle, B.IMAGE_COMPRESS_AUTO_GLZ, True, 36)
    assert seen.get("ZLIB_GLZ_RGB", 0) >= 3 and seen.get("raw", 0) >= 1
    assert seen.get("JPEG", 0) + seen.get("JPEG_ALPHA", 0) + seen.get("QUIC", 0) >= 2


@pytest.mark.gpu
def test_gpu_c_host_stream(tmp_path, gpu, oracle):
    exe = os.path.join(HOST, "host_cli")
    assert os.path.exists(exe), "build spice_amd/host first (make -C spice_amd/host)"
    seen = run_cli(exe, tmp_path, oracle, B.IMAGE_COMPRESS_AUTO_GLZ, True, 48)
    assert seen.get("ZLIB_GLZ_RGB", 0) + seen.get("GLZ_RGB", 0) >= 12
