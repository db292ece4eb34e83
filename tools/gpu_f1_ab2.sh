#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/f1ab; rm -rf $O; mkdir -p $O
for v in default "$@"; do
  if [ "$v" != default ]; then export SGPR_HIP_LIB=$R/variants/libsgpr_$v.so; else unset SGPR_HIP_LIB; fi
  for w in kitti world; do
  ( cd /tmp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_${v}_$w -o kt -- python $R/tools/f1_phases.py $w > $O/run_${v}_$w.log 2>&1 </dev/null )
  echo "== $v $w"; python $R/tools/kstats.py $(find $O/kt_${v}_$w -name kt_kernel_stats.csv | head -1) | grep "f1_scan\|f1_refine\|f1_plan\|slab_sum\|f1_final\|fill"
  done
done
rm -rf $O
