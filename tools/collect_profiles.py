#!/usr/bin/env python3
"""Turn gpurun_out/prof (tools/refresh_profiles.sh) into the committed summaries under profiles/.

usage: python tools/collect_profiles.py rNN"""
import collections
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1]
src, dst = "gpurun_out/prof", "profiles"


def pmc(name):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(set)
    for r in csv.DictReader(open(os.path.join(src, name + "_counter_collection.csv"))):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k].add(r["Dispatch_Id"])
    return {k: {c: v / len(calls[k]) for c, v in d.items()} for k, d in agg.items() if "sgpr" in k}


shutil.copy(os.path.join(src, "kt_kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats.csv"))
with open(os.path.join(dst, tag + "_kernel_stats.txt"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-cpu-baseline  (durations in us)\n")
    for r in csv.DictReader(open(os.path.join(src, "kt_kernel_stats.csv"))):
        f.write("%-100s calls %4s  avg %9.1f  min %9.1f  max %9.1f  %5s%%\n" % (
            r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3,
            r["Percentage"]))
for name in ("bench.json", "bench_under_rocprof.json"):
    line = open(os.path.join(src, name)).read().strip().splitlines()[-1]
    json.loads(line)
    open(os.path.join(dst, tag + "_" + name), "w").write(line + "\n")

fetch, write = pmc("fetch"), pmc("write")
hbm = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) -- python "
                 "tools/run_embed.py kitti00 3; per-launch averages (" + tag + ")",
       "unit_note": "FETCH_SIZE / WRITE_SIZE are KiB. WRITE_SIZE matches the known output bytes (pooled: 4541 x 128 B = "
                    "567.6 KiB). FETCH_SIZE under-reports on gfx950 (MI355X_MICROARCH.md: exactly 1/2 for 16 B/lane "
                    "streams); the corrected read traffic is bracketed by [raw, 2 x raw] - bench.py uses 2 x raw."}
for k in fetch:
    e = {"FETCH_SIZE_KiB": round(fetch[k].get("FETCH_SIZE", 0.0), 1), "WRITE_SIZE_KiB": round(write[k].get("WRITE_SIZE", 0.0), 1)}
    key = k.split("<")[0]
    if "embed" in k:
        e.update({"graphs_per_launch": 4541, "node_num": 100, "K": 10})
    hbm[key] = e
json.dump(hbm, open(os.path.join(dst, "pmc_hbm_latest.json"), "w"), indent=2)
with open(os.path.join(dst, tag + "_pmc_sq.txt"), "w") as f:
    f.write("# rocprofv3 --pmc (separate passes) on tools/run_embed.py kitti00 3; per-launch averages in millions (" + tag + ")\n")
    for name in ("sq1", "sq2"):
        for k, d in pmc(name).items():
            f.write("%-36s %s\n" % (k[:36], {c: round(v / 1e6, 3) for c, v in d.items()}))
print(open(os.path.join(dst, tag + "_kernel_stats.txt")).read())
print(json.dumps(hbm, indent=1)[:900])
print(open(os.path.join(dst, tag + "_pmc_sq.txt")).read())
