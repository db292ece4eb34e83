#!/bin/bash
# round 6, pass g: auto launches (oversize graphs inside the second pass), sgpr_size_order, then the whole GPU suite
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6g; rm -rf $O; mkdir -p $O
cd $R
( timeout 600 python -m pytest tests -m gpu -x -q -k "plain_embed or size_order" ) > $O/pytest_new.log 2>&1
tail -15 $O/pytest_new.log
python tools/run_auto.py 50 2>&1 | tee $O/auto_default.txt
( SGPR_SEQ_PARITY_OUT=$O/seq_parity_world.txt timeout 1700 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
