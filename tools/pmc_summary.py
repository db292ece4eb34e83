#!/usr/bin/env python3
"""Per-kernel, per-launch averages of rocprofv3 --pmc passes:  python tools/pmc_summary.py <dir> <pass name> ..."""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
for name in sys.argv[2:]:
    files = glob.glob(os.path.join(d, "**", name + "_counter_collection.csv"), recursive=True)
    if not files:
        print(name, ": no counter file")
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    for r in csv.DictReader(open(files[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add(r["Dispatch_Id"])
    for k, c in agg.items():
        if "sgpr" in k:
            print("%-8s %-40s %s" % (name, k[:40], {cn: round(v / len(calls[k]) / 1e6, 3) for cn, v in sorted(c.items())}))
