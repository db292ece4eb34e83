#!/bin/bash
# round 4, pass g: non-temporal stores in the all-pairs tail - time (kernel trace) and HBM write traffic (PMC), same box
set -x
export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/r4g; mkdir -p $O
cd /tmp
for v in default ntstore; do
  if [ "$v" != default ]; then export SGPR_HIP_LIB=$R/variants/libsgpr_$v.so; else unset SGPR_HIP_LIB; fi
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$v -o kt -- python $R/bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-end-to-end > $O/bench_$v.json 2> $O/bench_$v.err </dev/null
  echo "== $v"; python $R/tools/kstats.py $(find $O/kt_$v -name kt_kernel_stats.csv | head -1) | head -4
  for c in WRITE_SIZE FETCH_SIZE "TCC_WRITEBACK_sum TCC_WRITE_sum"; do
    n=$(echo $c | tr ' ' '_')
    timeout 100 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O -o p_${v}_$n -- python $R/tools/run_embed.py kitti00 3 > $O/p_${v}_$n.log 2>&1 </dev/null
  done
done
cd $R
python - <<'PY'
import csv,glob,collections,os
for f in sorted(glob.glob('gpurun_out/r4g/**/p_*_counter_collection.csv', recursive=True)):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:40]
        if 'score_all' in k or 'embed_kernel' in k or 'embed_redo' in k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,d in agg.items():
        print(os.path.basename(f)[:44], k, {c: round(sum(v)/len(v),1) for c,v in d.items()})
PY
