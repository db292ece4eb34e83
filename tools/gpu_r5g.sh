#!/bin/bash
# round 5, pass g: the any-shape kernels (larger architectures, node_num > 256, K > 32) against the oracle, then the whole suite
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5g; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "any_shape or beyond_the_tuned or smaller_architectures" > $O/anyshape.log 2>&1
tail -30 $O/anyshape.log
timeout 1500 python -m pytest tests -m gpu -q  > $O/gpu_tests.log 2>&1
tail -15 $O/gpu_tests.log
