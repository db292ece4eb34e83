#!/bin/bash
# round 5, second GPU pass: the whole GPU suite (no -x), then same-box A/B of the embed changes WITHOUT the fused tail
# preparation (so that the embed kernel alone is compared), the fused step against the two-launch step, NI = 2 in the tail.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5b; rm -rf $O; mkdir -p $O
cd $R
( SGPR_SEQ_PARITY_OUT=$O/seq_parity_world.txt timeout 2400 python -m pytest tests -m gpu -q -s ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
grep -E "^FAILED|^ERROR|config 5 full size|pair list 02|proven ties|random shapes|  pair [0-9]+ " $O/pytest.log | cut -c1-600
bench() {   # name, extra bench flags...; SGPR_HIP_LIB selects the library
  local name=$1; shift
  ( cd /tmp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$name -o kt -- python $R/bench.py --steps ${STEPS:-60} --warmup 5 --no-cpu-baseline --no-end-to-end --no-wide-range "$@" > $O/bench_$name.json 2> $O/bench_$name.err </dev/null )
  echo "== $name: $(python -c "import json,sys; r=json.loads([l for l in open('$O/bench_$name.json') if l.startswith('{')][-1]); print('step %.4f ms  embed %.4f  tail %s' % (r['ms_per_step'], r['kernel_durations']['embed_call_ms'], r['kernel_durations']['tail_call_ms']))" 2>&1 | tail -1)"
  python tools/kstats.py $(find $O/kt_$name -name kt_kernel_stats.csv | head -1) | head -${HEAD:-4}
}
unset SGPR_HIP_LIB
bench default
bench default_nofuse --no-fused-prep
for v in nostage nowpre noatt r4like; do
  if [ -f $R/variants/libsgpr_$v.so ]; then SGPR_HIP_LIB=$R/variants/libsgpr_$v.so bench ${v}_nofuse --no-fused-prep; fi
done
if [ -f $R/variants/libsgpr_ni2.so ]; then SGPR_HIP_LIB=$R/variants/libsgpr_ni2.so bench ni2; fi
unset SGPR_HIP_LIB
bench default_again
bench default_nofuse_again --no-fused-prep
