#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/f1s; rm -rf $O; mkdir -p $O
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt -- python $R/tools/run_f1.py 20 > $O/run.log 2>&1 </dev/null
cd $R; python tools/kstats.py $(find $O -name kt_kernel_stats.csv | head -1) | head -16
grep -v amdgpu $O/run.log | head
