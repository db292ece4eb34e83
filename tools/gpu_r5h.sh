#!/bin/bash
# round 5, pass h: the any-shape kernels after the channel-major / hoisted-tail rewrite: parity, then what they cost
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5h; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "any_shape or beyond_the_tuned or smaller_architectures or error_codes or test_gpu_modules" > $O/anyshape.log 2>&1
tail -30 $O/anyshape.log
timeout 500 python tools/run_anyshape.py $O/any_shape.txt 2>&1 | tail -12
