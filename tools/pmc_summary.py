"""Per-kernel averages of a rocprofv3 --pmc counter_collection.csv:
    python tools/pmc_summary.py FILE.csv"""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
first = None
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:44]
    first = first or r["Counter_Name"]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == first:
        n[k] += 1
for k in acc:
    print(k, n[k], {c: round(v / max(1, n[k])) for c, v in acc[k].items()})
