export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/tld; rm -rf $O; mkdir -p $O
cd /tmp
for ld in 4541 4544 4608; do
for c in FETCH_SIZE WRITE_SIZE; do
timeout 100 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O -o p_${c}_$ld -- python $R/tools/exp/tail_ld.py $ld > $O/p_${c}_$ld.log 2>&1 </dev/null
done; done
cd $R
python - <<'PY'
import csv,glob,collections
for ld in (4541,4544,4608):
  for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob('gpurun_out/tld/**/p_%s_%d_counter_collection.csv'%(c,ld), recursive=True)[0]
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:40]].append(float(r["Counter_Value"]))
    kt=glob.glob('gpurun_out/tld/**/p_%s_%d_kernel_trace.csv'%(c,ld), recursive=True)[0]
    d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(kt)) if 'score_all' in r['Kernel_Name']]
    for k,v in agg.items():
        if 'score_all' in k: print(ld, c, round(sum(v)/len(v),1), 'KiB', 'us', [round(x,1) for x in d])
PY
