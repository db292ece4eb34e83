#!/bin/bash
# round 5, fifth GPU pass: this round's library against round 4's on the same box, then the whole GPU suite
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5e; rm -rf $O; mkdir -p $O
cd $R
bench() {
  local name=$1; shift
  ( cd /tmp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$name -o kt -- python $R/bench.py --steps ${STEPS:-60} --warmup 5 --no-cpu-baseline --no-end-to-end --no-wide-range "$@" > $O/bench_$name.json 2> $O/bench_$name.err </dev/null )
  echo "== $name: $(python -c "import json,sys; r=json.loads([l for l in open('$O/bench_$name.json') if l.startswith('{')][-1]); print('step %.4f ms  embed %.4f  tail %s' % (r['ms_per_step'], r['kernel_durations']['embed_call_ms'], r['kernel_durations']['tail_call_ms']))" 2>&1 | tail -1)"
  python tools/kstats.py $(find $O/kt_$name -name kt_kernel_stats.csv | head -1) | head -${HEAD:-4}
}
unset SGPR_HIP_LIB
bench default
SGPR_HIP_LIB=$R/variants/libsgpr_r4.so bench r4
SGPR_HIP_LIB=$R/variants/libsgpr_oldpro.so bench oldpro
unset SGPR_HIP_LIB
bench default_again
SGPR_HIP_LIB=$R/variants/libsgpr_r4.so bench r4_again
unset SGPR_HIP_LIB
STEPS=200 bench pairs128 --workload pairs128
SGPR_HIP_LIB=$R/variants/libsgpr_r4.so STEPS=200 bench pairs128_r4 --workload pairs128
unset SGPR_HIP_LIB
( SGPR_SEQ_PARITY_OUT=$O/seq_parity_world.txt timeout 2400 python -m pytest tests -m gpu -q -s ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
grep -E "^FAILED|^ERROR" $O/pytest.log | cut -c1-300
