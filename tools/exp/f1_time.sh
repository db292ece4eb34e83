export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/f1; rm -rf $O; mkdir -p $O
timeout 300 python tools/run_f1.py 10 check > $O/run_f1.log 2>&1; cat $O/run_f1.log
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt -- python $GRAFT_REPO_ROOT/tools/run_f1.py 3 > $O/kt.log 2>&1 </dev/null
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/f1/**/kt_kernel_stats.csv', recursive=True)
for r in csv.DictReader(open(f[0])):
    print(r['Name'][:70], r['Calls'], r['AverageNs'], r['Percentage'])
PY
