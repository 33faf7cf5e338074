"""Per-launch HBM traffic of the dominant kernels from two rocprofv3 PMC passes
(FETCH_SIZE, WRITE_SIZE), written into profiles/traffic.json.

usage: pmc_traffic.py FETCH_counter_collection.csv WRITE_counter_collection.csv WARMUP SESSIONS POOL FLIGHTS FRAMES
Counter values are KiB per XCD instance and are summed per dispatch; the first
WARMUP x FLIGHTS dispatches of the GLZ kernel are dropped (bench.py's untimed steps; with
FLIGHTS batches in flight a step is FLIGHTS launches of SESSIONS / FLIGHTS images each, and
the figures are per launch).  A csv of a run without one of the kernels leaves that kernel's
entry alone."""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = {"glz_parse_kernel": "glz1080", "jpeg_strip_kernel": "mjpeg2160"}


def per_launch(path, counter, warmup):
    by = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        for k in KERNELS:
            if k in r["Kernel_Name"]:
                by[k][int(r["Dispatch_Id"])] += float(r["Counter_Value"]) * 1024.0
    out = {}
    for k, d in by.items():
        vals = [d[i] for i in sorted(d)][warmup if k == "glz_parse_kernel" else 0:]
        if not vals:
            continue
        out[k] = sum(vals) / len(vals)
    return out


flights = int(sys.argv[6]) if len(sys.argv) > 6 else 1
fetch = per_launch(sys.argv[1], "FETCH_SIZE", int(sys.argv[3]) * flights)
write = per_launch(sys.argv[2], "WRITE_SIZE", int(sys.argv[3]) * flights)
path = os.path.join(ROOT, "profiles", "traffic.json")
doc = json.load(open(path))
for k, name in KERNELS.items():
    if k in fetch:
        doc[name].update(fetch_bytes_raw=fetch[k], write_bytes=write[k],
                         traffic_bytes=int(2 * fetch[k] + write[k]))
if "glz_parse_kernel" in fetch:
    doc["glz1080"]["sessions_per_gpu"] = int(sys.argv[4])
    doc["glz1080"]["distinct_sequences_per_gpu"] = int(sys.argv[5])
    doc["glz1080"]["batches_in_flight"] = flights
if "jpeg_strip_kernel" in fetch and len(sys.argv) > 7:
    doc["mjpeg2160"]["frames_per_step_per_gpu"] = int(sys.argv[7])
    doc["mjpeg2160"]["distinct_frames_per_gpu"] = int(sys.argv[7])
json.dump(doc, open(path, "w"), indent=1)
print(json.dumps(doc, indent=1))
