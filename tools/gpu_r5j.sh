#!/bin/bash
# round 5, pass j: the multi-rectangle tail (config 4) at other occupancies / with two row graphs interleaved - same box.
# The variants are builds (tools/build_variant.sh m_ni<N>_occ<O> --src <copy>) of a copy of the tree in which the two
# constants of launch_score_all_pairs_multi - AP_MULTI_OCC and the kernel's NI template argument - were edited by hand.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5j; rm -rf $O; mkdir -p $O
cd $R
for v in base m_ni2_occ3 m_ni1_occ3 m_ni2_occ2 base; do
  if [ $v == base ]; then unset SGPR_HIP_LIB; else export SGPR_HIP_LIB=$R/variants/libsgpr_$v.so; fi
  timeout 300 python bench.py --workload kitti5seq --no-cpu-baseline --no-end-to-end --steps 50 > $O/bench_$v.json 2> $O/bench_$v.err
  python - <<PY
import json
r = json.loads([l for l in open("$O/bench_$v.json") if l.startswith("{")][-1])
print("$v", "step %.4f ms" % r["ms_per_step"], r["kernel_durations"])
PY
done
export SGPR_HIP_LIB=$R/variants/libsgpr_m_ni2_occ3.so
timeout 600 python -m pytest tests -m gpu -x -q -k "sequence_set or multi or shard_invariance or all_pairs_matrix" > $O/pytest_ni2occ3.log 2>&1; tail -3 $O/pytest_ni2occ3.log
