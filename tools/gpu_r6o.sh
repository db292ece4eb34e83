#!/bin/bash
# round 6, pass o: the matrix-core any-shape embed (sgpr_wide.hip): the architecture tests, then its time next to the plain-fp32 kernel
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6o; rm -rf $O; mkdir -p $O
cd $R
( timeout 1200 python -m pytest tests -m gpu -x -q -k "larger_architectures or smaller_architectures or node_num_and_k_beyond or custom_ops" ) > $O/pytest.log 2>&1; tail -15 $O/pytest.log | cut -c1-300
timeout 600 python tools/run_anyshape.py $O/any_shape.txt 2>&1 | grep -v amdgpu.ids | cut -c1-400
