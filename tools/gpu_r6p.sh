#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6p; rm -rf $O; mkdir -p $O
cd $R
( timeout 1200 python -m pytest tests -m gpu -x -q -k "matrix_core_any_shape or larger_architectures" ) > $O/pytest.log 2>&1; tail -12 $O/pytest.log | cut -c1-400
