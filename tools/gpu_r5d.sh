#!/bin/bash
# round 5, fourth GPU pass: the self-preparing tail (sgpr_embed_ex + sgpr_score_all_pairs_prepared, per-workgroup-local
# operands) against the two-launch tail on the same box, then the whole GPU suite
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5d; rm -rf $O; mkdir -p $O
cd $R
( timeout 600 python -m pytest tests -m gpu -q -x -k "tail_operands or f16_range_guard or all_pairs_matrix or shard_invariance or scorer_on_a_ragged or sequence_set" ) > $O/pytest_quick.log 2>&1
tail -3 $O/pytest_quick.log
bench() {
  local name=$1; shift
  ( cd /tmp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$name -o kt -- python $R/bench.py --steps ${STEPS:-60} --warmup 5 --no-cpu-baseline --no-end-to-end --no-wide-range "$@" > $O/bench_$name.json 2> $O/bench_$name.err </dev/null )
  echo "== $name: $(python -c "import json,sys; r=json.loads([l for l in open('$O/bench_$name.json') if l.startswith('{')][-1]); print('step %.4f ms  embed %.4f  tail %s' % (r['ms_per_step'], r['kernel_durations']['embed_call_ms'], r['kernel_durations']['tail_call_ms']))" 2>&1 | tail -1)"
  python tools/kstats.py $(find $O/kt_$name -name kt_kernel_stats.csv | head -1) | head -${HEAD:-5}
}
unset SGPR_HIP_LIB
bench default_nofuse --no-fused-prep
bench default
bench default_nofuse_again --no-fused-prep
bench default_again
( SGPR_SEQ_PARITY_OUT=$O/seq_parity_world.txt timeout 2400 python -m pytest tests -m gpu -q -s ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
grep -E "^FAILED|^ERROR|config 5 full size|pair list 02|proven ties|random shapes|  pair [0-9]+ " $O/pytest.log | cut -c1-400
