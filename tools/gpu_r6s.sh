#!/bin/bash
# round 6, pass s: the matrix-core any-shape all-pairs tail (sgpr_wide.hip): tests, run_anyshape, kernel trace of the tail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6s; rm -rf $O; mkdir -p $O
cd $R
( timeout 1200 python -m pytest tests -m gpu -x -q -k "matrix_core_any_shape or larger_architectures or any_shape" ) > $O/pytest.log 2>&1; tail -15 $O/pytest.log | cut -c1-400
( timeout 900 python tools/run_anyshape.py $O/any_shape.txt ) > $O/anyshape.log 2>&1; tail -3 $O/anyshape.log | cut -c1-300
grep -E "all-pairs|same handle's tail" $O/any_shape.txt | cut -c1-600
( cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $R/tools/run_anyshape.py > $O/run_kt.log 2>&1 </dev/null )
python tools/kstats.py $(find $O/kt -name kt_kernel_stats.csv | head -1) | grep -E "tail|rect|score" | head
