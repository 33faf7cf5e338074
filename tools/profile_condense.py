"""Condense what tools/profile_round.sh left under gpurun_out/prof_TAG/ into the files kept
under profiles/:
    python tools/profile_condense.py TAG [GLZ_NAME]
kernel stats and the bench line of the same run per workload, one CSV of the FETCH_SIZE /
WRITE_SIZE dispatches (summed over the XCDs, in the counter's own unit), one CSV per workload
of the SQ counters per dispatch."""
import collections
import csv
import glob
import os
import shutil
import sys

tag = sys.argv[1]
glz_name = sys.argv[2] if len(sys.argv) > 2 else "glz1080_s3072"
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
names = {"glz1080": glz_name, "mjpeg2160": "mjpeg2160"}


def one(pattern):
    hits = glob.glob(os.path.join(src, pattern))
    assert len(hits) == 1, (pattern, hits)
    return hits[0]


def short(kernel):
    return kernel.split("(")[0]


def ours(kernel):
    """the library's own kernels, not the runtime's fills and copies"""
    return any(t in kernel for t in ("glz_", "jpeg_", "gradual_"))


for w, name in names.items():
    shutil.copy(one("stats_%s/*/*_kernel_stats.csv" % w),
                os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, name)))
    line = [ln for ln in open(os.path.join(src, w + "_line.log")) if ln.startswith("{")][-1]
    open(os.path.join(dst, "%s_%s_line.json" % (tag, name)), "w").write(line)
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(one("pmc_sq_%s/*/*_counter_collection.csv" % w))):
        if not ours(r["Kernel_Name"]):
            continue
        key = (int(r["Dispatch_Id"]), short(r["Kernel_Name"]), r["Counter_Name"])
        acc[key] = acc.get(key, 0.0) + float(r["Counter_Value"])
    with open(os.path.join(dst, "%s_%s_pmc_sq.csv" % (tag, name)), "w") as f:
        f.write("dispatch_id,kernel,counter,value\n")
        for (d, k, c), v in sorted(acc.items()):
            f.write("%d,%s,%s,%d\n" % (d, k.replace("void ", ""), c, v))

# the kernel for an under-filled GPU, when the round's script profiled it
hits = glob.glob(os.path.join(src, "stats_sessions64/*/*_kernel_stats.csv"))
if hits:
    shutil.copy(hits[0], os.path.join(dst, "%s_sessions64_kernel_stats.csv" % tag))
    line = [ln for ln in open(os.path.join(src, "sessions64_line.log")) if ln.startswith("{")][-1]
    open(os.path.join(dst, "%s_sessions64_line.json" % tag), "w").write(line)

with open(os.path.join(dst, "%s_pmc_hbm.csv" % tag), "w") as f:
    f.write("pass,dispatch_id,kernel,counter,KiB_summed_over_XCDs\n")
    for w, p in (("glz1080", "g"), ("mjpeg2160", "m")):
        for c, q in (("FETCH_SIZE", "f"), ("WRITE_SIZE", "w")):
            acc = collections.OrderedDict()
            for r in csv.DictReader(open(one("pmc_%s_%s/*/*_counter_collection.csv" % (c, w)))):
                if r["Counter_Name"] != c or not ours(r["Kernel_Name"]):
                    continue
                key = (int(r["Dispatch_Id"]), short(r["Kernel_Name"]))
                acc[key] = acc.get(key, 0.0) + float(r["Counter_Value"])
            for (d, k), v in sorted(acc.items()):
                f.write("%s%s,%d,%s,%s,%d\n" % (p, q, d, k, c, v))
print("condensed", tag)
