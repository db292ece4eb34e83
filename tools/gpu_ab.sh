#!/bin/bash
# same-box A/B: bash tools/gpu_ab.sh "<pytest -k expression or empty>" variant ...   (default library first)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ab; rm -rf $O; mkdir -p $O
cd $R
K="$1"; shift
if [ -n "$K" ]; then ( timeout 900 python -m pytest tests -m gpu -x -q -k "$K" ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log; fi
for v in default "$@"; do
  if [ "$v" != default ]; then export SGPR_HIP_LIB=$R/variants/libsgpr_$v.so; else unset SGPR_HIP_LIB; fi
  for w in ${WORKLOADS:-kitti00}; do
    ( cd /tmp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_${v}_$w -o kt -- python $R/bench.py --workload $w --steps ${STEPS:-60} --warmup 5 --no-cpu-baseline --no-end-to-end > $O/bench_${v}_$w.json 2> $O/bench_${v}_$w.err </dev/null )
    echo "== $v $w"; python tools/kstats.py $(find $O/kt_${v}_$w -name kt_kernel_stats.csv | head -1) | head -${HEAD:-2}
  done
done
