# per-phase instruction counts of the embed kernel: one rocprofv3 --pmc pass per ablation mask
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmca; rm -rf $O; mkdir -p $O
cd /tmp
for m in 256 257 258 260 264 271 16655 33039 49423 288 272; do
  timeout 100 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O -o m$m -- python $GRAFT_REPO_ROOT/tools/run_embed.py kitti00 2 $m > $O/m$m.log 2>&1 </dev/null
done
cd $GRAFT_REPO_ROOT
for m in 256 257 258 260 264 271 16655 33039 49423 288 272; do python tools/pmc_summary.py $O m$m | grep embed; done
python - <<'PY'
import csv,glob,collections
for m in [256,257,258,260,264,271,272,288]:
    f=glob.glob('gpurun_out/pmca/**/m%d_kernel_trace.csv'%m, recursive=True)
    if not f: continue
    d=[ (float(r['End_Timestamp'])-float(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(f[0])) if 'embed' in r['Kernel_Name']]
    print('mask',m,'embed us',[round(x,1) for x in d])
PY
