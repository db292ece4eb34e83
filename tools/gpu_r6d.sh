#!/bin/bash
# round 6, pass d: tests of the big instance + same-box A/B of variants on the stress shape:  bash tools/gpu_r6d.sh variant ...
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6d; rm -rf $O; mkdir -p $O
cd $R
( timeout 1500 python -m pytest tests -m gpu -x -q -k "${TESTS:-config5 or odd_sizes or synthetic_golden or random_shapes or label_lookup or ordered_embed or node_cap or stress_shape or lean_plans or f16_planes_range or ragged}" ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
for v in default "$@"; do
  if [ "$v" != default ]; then export SGPR_HIP_LIB=$R/variants/libsgpr_$v.so; else unset SGPR_HIP_LIB; fi
  ( cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_${v} -o kt -- python $R/tools/run_embed.py ${SHAPE:-stress} 30 > $O/run_${v}.log 2>&1 </dev/null )
  echo "== $v"; python tools/kstats.py $(find $O/kt_${v} -name kt_kernel_stats.csv | head -1) | grep embed | head -1
done
