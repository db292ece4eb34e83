#!/usr/bin/env python3
"""Per (kernel, workgroup size, grid) average duration and start-offset within a step, from *_kernel_trace.csv."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/kt/kt_kernel_trace.csv")))
agg = collections.defaultdict(list)
for r in rows:
    key = (r["Kernel_Name"][:36], r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")), r.get("Grid_Size_X", r.get("Grid_Size", "?")),
           r.get("LDS_Block_Size", "?"), r.get("Queue_Id", "?"))
    agg[key].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for k, v in agg.items():
    d = [e - s for s, e in v]
    print("%-38s wg %4s grid %7s lds %6s q %s  n %3d  avg %8.1f us" % (k + (len(v), sum(d) / len(d) / 1e3)))
# timeline of the last 8 launches
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[-8]["Start_Timestamp"])
for r in rows[-8:]:
    print("%-30s wg %4s  start %8.1f  end %8.1f us" % (r["Kernel_Name"][:30], r.get("Workgroup_Size_X", "?"),
          (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3))
