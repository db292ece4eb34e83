#!/bin/bash
# round 6, pass e: plain sgpr_embed (auto lean plan + hand-over) against capped / ordered launches, with and without the auto path
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6e; rm -rf $O; mkdir -p $O
cd $R
python tools/run_auto.py 50 2>&1 | tee $O/auto_default.txt
SGPR_HIP_LIB=$R/variants/libsgpr_noauto.so python tools/run_auto.py 50 2>&1 | tee $O/auto_noauto.txt
( timeout 1500 python -m pytest tests -m gpu -x -q -k "${TESTS:-node_cap or ordered_embed or error_codes or odd_sizes or shard_invariance or label_lookup or ragged or super_node or lean_plans or split_launch}" ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
