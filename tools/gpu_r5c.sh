#!/bin/bash
# round 5, third GPU pass: where the fused tail preparation's time goes (timing-only arrival variants), the merged
# prologue / layer-3 prefetch build against the library of the previous pass
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5c; rm -rf $O; mkdir -p $O
cd $R
( timeout 600 python -m pytest tests -m gpu -q -x -k "tail_operands or shipped_graphs or synthetic_golden or super_node or label_lookup or lean_plans or odd_sizes or ragged_store or generic_branch or split_launch" ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
bench() {
  local name=$1; shift
  ( cd /tmp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$name -o kt -- python $R/bench.py --steps ${STEPS:-60} --warmup 5 --no-cpu-baseline --no-end-to-end --no-wide-range "$@" > $O/bench_$name.json 2> $O/bench_$name.err </dev/null )
  echo "== $name: $(python -c "import json,sys; r=json.loads([l for l in open('$O/bench_$name.json') if l.startswith('{')][-1]); print('step %.4f ms  embed %.4f  tail %s' % (r['ms_per_step'], r['kernel_durations']['embed_call_ms'], r['kernel_durations']['tail_call_ms']))" 2>&1 | tail -1)"
  python tools/kstats.py $(find $O/kt_$name -name kt_kernel_stats.csv | head -1) | head -${HEAD:-4}
}
unset SGPR_HIP_LIB
bench default_nofuse --no-fused-prep
bench default
for v in arr1 arr2 arr3; do
  if [ -f $R/variants/libsgpr_$v.so ]; then SGPR_HIP_LIB=$R/variants/libsgpr_$v.so bench $v; fi
done
unset SGPR_HIP_LIB
bench default_nofuse_again --no-fused-prep
