#!/bin/bash
# round 6, pass a: the owned-rows instance beyond 64 slots (embed_big_kernel) - the tests that exercise it, then a same-box
# A/B against the chunked plans (variants/libsgpr_nobig.so = -DSGPR_BIG_OWNED=0) on stress / kitti00-uncapped shapes.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6a; rm -rf $O; mkdir -p $O
cd $R
( timeout 1500 python -m pytest tests -m gpu -x -q -k "config5 or odd_sizes or synthetic_golden or random_shapes or label_lookup or ordered_embed or node_cap or stress_shape or lean_plans or f16_planes_range or ragged" ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
for v in default nobig; do
  if [ "$v" != default ]; then export SGPR_HIP_LIB=$R/variants/libsgpr_$v.so; else unset SGPR_HIP_LIB; fi
  ( cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_${v}_stress -o kt -- python $R/bench.py --workload stress --steps 100 --warmup 5 --no-cpu-baseline --no-end-to-end > $O/bench_${v}_stress.json 2> $O/bench_${v}_stress.err </dev/null )
  echo "== $v stress"; python tools/kstats.py $(find $O/kt_${v}_stress -name kt_kernel_stats.csv | head -1) | head -3
done
unset SGPR_HIP_LIB
