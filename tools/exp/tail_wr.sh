#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/twr; rm -rf $O; mkdir -p $O
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o -E "TCC_EA0?_WR[A-Z0-9_]*|TCC_EA0?_RD[A-Z0-9_]*|TCC_WRITEBACK[A-Z_]*|TCC_NORMAL_WRITEBACK[A-Z_]*|TCC_ALL_TC_OP_WB[A-Z_]*|TCC_WRITE[A-Z_]*|TCC_ATOMIC[A-Z_]*" | sort -u | tr '\n' ' ' > $O/counters.txt
cat $O/counters.txt; echo
for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_WRITE_sum TCC_WRITEBACK_sum" "TCC_NORMAL_WRITEBACK_sum TCC_ALL_TC_OP_WB_WRITEBACK_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum"; do
  n=$(echo $set | tr ' ' '_')
  timeout 100 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O -o t_$n -- python $R/tools/exp/tail_ld.py 4544 > $O/t_$n.log 2>&1 </dev/null
  timeout 100 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O -o c_$n -- $R/tools/probes/store_calib_probe > $O/c_$n.log 2>&1 </dev/null
done
cd $R
python - <<'PY'
import csv, glob, os, collections
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/twr")
for f in sorted(glob.glob(os.path.join(O, "**", "*_counter_collection.csv"), recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:34]
        if "score_all" in k or "store_kernel<0>" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in agg.items():
        print(os.path.basename(f)[:40], k, {c: round(sum(v) / len(v) / 1e3, 1) for c, v in d.items()}, "(thousands per launch)")
PY
