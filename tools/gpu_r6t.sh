#!/bin/bash
# round 6, pass t: sgpr_f1_max - tests, phases, A/B of the current sources against the variants named in $VARIANTS
# (tools/build_variant.sh; e.g. nocull = -DSGPR_F1_CULL=0, sh18 = -DSGPR_F1_SH=18), kernel trace
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6t; rm -rf $O; mkdir -p $O
cd $R
( timeout 1500 python -m pytest tests -m gpu -x -q -k "f1 or counts or metrics or sequence or consumers" ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log | cut -c1-300
for v in default ${VARIANTS} default ${VARIANTS}; do
  if [ "$v" != default ]; then export SGPR_HIP_LIB=$R/variants/libsgpr_$v.so; else unset SGPR_HIP_LIB; fi
  for k in kitti world; do echo "== $v $k"; timeout 300 python tools/f1_phases.py $k 2>&1 | grep -v amdgpu.ids | head -2 | cut -c1-260; done
done
for v in default ${TRACE_VARIANTS}; do
  if [ "$v" != default ]; then export SGPR_HIP_LIB=$R/variants/libsgpr_$v.so; else unset SGPR_HIP_LIB; fi
  ( cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$v -o kt -- python $R/tools/f1_phases.py kitti 20 > $O/run_kt_$v.log 2>&1 </dev/null )
  echo "== trace $v"; python tools/kstats.py $(find $O/kt_$v -name kt_kernel_stats.csv | head -1) | grep -E "f1_|slab" | head
done
if [ -n "$TEST_VARIANT" ]; then
  export SGPR_HIP_LIB=$R/variants/libsgpr_$TEST_VARIANT.so
  ( timeout 1500 python -m pytest tests -m gpu -x -q -k "f1 or counts or metrics or sequence or consumers" ) > $O/pytest_$TEST_VARIANT.log 2>&1; echo "== tests on $TEST_VARIANT"; tail -4 $O/pytest_$TEST_VARIANT.log | cut -c1-300
fi
