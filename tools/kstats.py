#!/usr/bin/env python3
"""Print name / calls / avg / min / max (us) from a rocprofv3 *_kernel_stats.csv."""
import csv
import sys

for r in csv.DictReader(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/kt/kt_kernel_stats.csv")):
    print("%-44s calls %4s  avg %9.1f  min %9.1f  max %9.1f us" % (
        r["Name"][:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
