#!/bin/bash
# round 6, pass n: the whole GPU suite on the final sources + the plain / default / ordered embed table
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6n; rm -rf $O; mkdir -p $O
cd $R
( SGPR_SEQ_PARITY_OUT=$O/seq_parity_world.txt timeout 2400 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log
python tools/run_auto.py 50 2>&1 | grep -v amdgpu.ids | tee $O/auto.txt | cut -c1-230
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
