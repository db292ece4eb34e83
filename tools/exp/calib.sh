#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/calib; rm -rf $O; mkdir -p $O
cd /tmp
timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O -o w -- $R/tools/probes/store_calib_probe > $O/calib_w.log 2>&1 </dev/null
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O -o f -- $R/tools/probes/store_calib_probe > $O/calib_f.log 2>&1 </dev/null
cd $R
python - <<'PY'
import csv, glob, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/calib")
names = ["tail16 ld=4541", "tail16 ld=4544 +4B", "tail16 ld=4544 aligned", "seg64 ld=4541", "row1k ld=4541", "tail16 ld=4541 delayed", "tail16 ld=4544 delayed"]
kt = glob.glob(os.path.join(O, "**", "w_kernel_trace.csv"), recursive=True)
dur = {}
if kt:
    for r in csv.DictReader(open(kt[0])):
        dur[r["Dispatch_Id"]] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
for tag in ("w", "f"):
    f = glob.glob(os.path.join(O, "**", tag + "_counter_collection.csv"), recursive=True)
    if not f:
        print(tag, "no counter file"); continue
    rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Dispatch_Id"]))
    for i, r in enumerate(rows[7:14]):       # second repetition
        print("%-26s %s %10.1f KiB  (x %.3f of 80549.2 KiB)  %s" % (names[i % 7], r["Counter_Name"], float(r["Counter_Value"]), float(r["Counter_Value"]) / 80549.2,
              ("%.1f us" % dur[r["Dispatch_Id"]]) if (tag == "w" and r["Dispatch_Id"] in dur) else ""))
PY
