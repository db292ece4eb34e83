#!/bin/bash
# round 5, sixth GPU pass: the L2 input prefetch (one round of resident workgroups ahead) on / off / at half the distance
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5f; rm -rf $O; mkdir -p $O
cd $R
( timeout 600 python -m pytest tests -m gpu -q -x -k "shipped_graphs or synthetic_golden or ordered_embed or node_cap or lean_plans or odd_sizes or ragged_store or generic_branch or split_launch or shard_invariance" ) > $O/pytest.log 2>&1
tail -2 $O/pytest.log
bench() {
  local name=$1; shift
  ( cd /tmp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$name -o kt -- python $R/bench.py --steps ${STEPS:-60} --warmup 5 --no-cpu-baseline --no-end-to-end --no-wide-range "$@" > $O/bench_$name.json 2> $O/bench_$name.err </dev/null )
  echo "== $name: $(python -c "import json,sys; r=json.loads([l for l in open('$O/bench_$name.json') if l.startswith('{')][-1]); print('step %.4f ms  embed %.4f  tail %s' % (r['ms_per_step'], r['kernel_durations']['embed_call_ms'], r['kernel_durations']['tail_call_ms']))" 2>&1 | tail -1)"
  python tools/kstats.py $(find $O/kt_$name -name kt_kernel_stats.csv | head -1) | head -${HEAD:-2}
}
unset SGPR_HIP_LIB
bench default
for v in pf0 pf512; do SGPR_HIP_LIB=$R/variants/libsgpr_$v.so bench $v; done
unset SGPR_HIP_LIB
bench default_again
SGPR_HIP_LIB=$R/variants/libsgpr_pf0.so bench pf0_again
unset SGPR_HIP_LIB
STEPS=40 bench kitti5seq --workload kitti5seq
SGPR_HIP_LIB=$R/variants/libsgpr_pf0.so STEPS=40 bench kitti5seq_pf0 --workload kitti5seq
unset SGPR_HIP_LIB
STEPS=100 bench pairlist --workload pairlist
SGPR_HIP_LIB=$R/variants/libsgpr_pf0.so STEPS=100 bench pairlist_pf0 --workload pairlist
