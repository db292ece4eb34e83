#!/bin/bash
# round-3 first GPU pass: tests, same-box A/B of library variants, per-phase LDS counters, WRITE_SIZE calibration
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3a; rm -rf $O; mkdir -p $O
cd $R
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1
tail -8 $O/pytest.log
for v in default r2 norot; do
  if [ "$v" != default ]; then export SGPR_HIP_LIB=$R/variants/libsgpr_$v.so; else unset SGPR_HIP_LIB; fi
  rm -rf $O/kt_$v
  ( cd /tmp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$v -o kt -- python $R/bench.py --steps 60 --warmup 5 --no-cpu-baseline --no-end-to-end > $O/bench_$v.json 2> $O/bench_$v.err </dev/null )
  echo "== $v"; python tools/kstats.py $(find $O/kt_$v -name kt_kernel_stats.csv | head -1) | head -4
done
unset SGPR_HIP_LIB
# per-phase LDS counters (ablation masks on the profile instance)
cd /tmp
for shape in kitti00 stress; do
  for m in 256 257 258 260 264 271 16655; do
    timeout 120 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS --output-format csv -d $O/lds -o ${shape}_m$m -- python $R/tools/run_embed.py $shape 2 $m > $O/lds_${shape}_m$m.log 2>&1 </dev/null
  done
done
cd $R
for shape in kitti00 stress; do for m in 256 257 258 260 264 271 16655; do python tools/pmc_summary.py $O/lds ${shape}_m$m | grep "embed_kernel"; done; done > $O/lds_summary.txt
cat $O/lds_summary.txt
# WRITE_SIZE / FETCH_SIZE calibration of the tail's store shapes
cd /tmp
timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/calib -o w -- $R/tools/probes/store_calib_probe > $O/calib_w.log 2>&1 </dev/null
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/calib -o f -- $R/tools/probes/store_calib_probe > $O/calib_f.log 2>&1 </dev/null
cd $R
python - <<'PY'
import csv, glob, os
O = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r3a/calib")
for tag in ("w", "f"):
    f = glob.glob(os.path.join(O, "**", tag + "_counter_collection.csv"), recursive=True)
    if not f:
        print(tag, "no counter file"); continue
    rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Dispatch_Id"]))
    print(tag, [(r["Dispatch_Id"], r["Counter_Name"], round(float(r["Counter_Value"]), 1)) for r in rows][:15])
PY
timeout 200 python bench.py --workload pairs128 --steps 200 --no-cpu-baseline > $O/bench_pairs128.json 2> $O/bench_pairs128.err; python -c "
import json,sys
for w in ('pairs128',):
    d=json.loads([l for l in open('$O/bench_%s.json'%w) if l.startswith('{')][-1]); print(w, d['ms_per_step'], d['roofline']['launch_ms'])
for v in ('default','r2','norot'):
    d=json.loads([l for l in open('$O/bench_%s.json'%v) if l.startswith('{')][-1]); print(v, 'ms/step', round(d['ms_per_step'],4), 'embed', round(d['roofline']['launch_ms'],4), 'tail', round(d['roofline_tail']['launch_ms'],4))
"
