#!/bin/bash
# same-box A/B of the one-call F1-max: bash tools/gpu_f1_ab.sh variant ...   (default library first)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/f1ab; rm -rf $O; mkdir -p $O
for v in default "$@"; do
  if [ "$v" != default ]; then export SGPR_HIP_LIB=$R/variants/libsgpr_$v.so; else unset SGPR_HIP_LIB; fi
  ( cd /tmp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$v -o kt -- python $R/tools/run_f1.py 3 check > $O/run_$v.log 2>&1 </dev/null )
  echo "== $v"; python $R/tools/kstats.py $(find $O/kt_$v -name kt_kernel_stats.csv | head -1) | grep "f1_scan\|f1_refine\|f1_plan"
  ( cd $R; for w in kitti world; do timeout 100 python tools/f1_phases.py $w 2>&1 | grep "per call\|phases" | head -2; done )
done
rm -rf $O
