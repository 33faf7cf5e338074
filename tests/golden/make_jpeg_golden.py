"""Generates tests/golden/jpeg_golden.json from the REAL libjpeg.

Run in the build container only (needs oracle/_ref/libjpegref.so, i.e. libjpeg
headers): the vectors are outputs of IJG libjpeg 9d driven with the reference's
call sequence (oracle/ref_jpeg_driver.c; server/mjpeg_encoder.c:771-823 for
IFAST frames, server/jpeg_encoder.c:232-246 for ISLOW stills), with
do_fancy_downsampling = FALSE (the v6b / libjpeg-turbo downsampler the
reference's distro libjpeg has).  Inputs come from spice_amd.frames.golden_image,
which depends on nothing but integer arithmetic.
"""
import base64
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from spice_amd import frames  # noqa: E402

CASES = []
for (w, h) in [(1, 1), (8, 8), (16, 16), (17, 9), (33, 17), (64, 48), (250, 130), (641, 33),
               (640, 480), (1280, 720)]:
    for kind in ("noise", "smooth", "desktop"):
        for (q, dct) in ((70, 1), (80, 1), (20, 1), (85, 0), (100, 0)):
            if w * h > 100000 and (q, dct) not in ((70, 1), (85, 0)):
                continue
            CASES.append((w, h, kind, q, dct))
CASES.append((3840, 2160, "smooth", 70, 1))
CASES.append((3840, 2160, "smooth", 85, 0))
CASES.append((1920, 1080, "desktop", 85, 0))
# 4K frames that take the long Huffman paths (text edges, noise), and the stream encoder's
# quality 80
CASES.append((3840, 2160, "desktop", 80, 1))
CASES.append((3840, 2160, "noise", 80, 1))
CASES.append((3840, 2160, "desktop", 85, 0))
CASES.append((3840, 2160, "noise", 70, 1))
CASES.append((3840, 2160, "smooth", 80, 1))


def pixels(fmt, bgrx):
    if fmt == O.FMT_32BIT:
        return bgrx.reshape(bgrx.shape[0], -1)
    if fmt == O.FMT_24BIT:
        return frames.to_bgr24(bgrx)
    return frames.to_rgb16(bgrx)


def main():
    out = {"libjpeg": "IJG libjpeg 9d (JPEG_LIB_VERSION 90), do_fancy_downsampling=FALSE",
           "cases": []}
    for (w, h, kind, q, dct) in CASES:
        seed = (w * 31 + h * 17 + q) & 0xffff
        img = frames.golden_image(w, h, kind, seed)
        fmts = [O.FMT_32BIT] if w * h > 5000 else [O.FMT_32BIT, O.FMT_24BIT, O.FMT_16BIT]
        for fmt in fmts:
            px = pixels(fmt, img)
            ref = O.ref_jpeg_encode(px, w, h, px.shape[1], fmt, q, dct, 0)
            rec = {"w": w, "h": h, "kind": kind, "seed": seed, "fmt": fmt, "q": q, "dct": dct,
                   "len": int(len(ref)), "sha256": hashlib.sha256(ref.tobytes()).hexdigest()}
            if w * h <= 300:
                rec["bytes"] = base64.b64encode(ref.tobytes()).decode()
            out["cases"].append(rec)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "jpeg_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", len(out["cases"]), "cases to", path)


if __name__ == "__main__":
    main()
