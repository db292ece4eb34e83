#!/bin/bash
# PC sampling of the embed kernel (rocprofv3 beta): bash tools/exp/pcsample.sh [host_trap|stochastic] [shape]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pcs; rm -rf $O; mkdir -p $O
M=${1:-host_trap}; S=${2:-kitti00}
cd /tmp
if [ "$M" == "stochastic" ]; then U=cycles; I=${3:-65536}; else U=time; I=${3:-1}; fi
timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $M --pc-sampling-unit $U --pc-sampling-interval $I --kernel-trace --output-format csv -d $O -o pcs -- python $R/tools/run_embed.py $S 40 > $O/run.log 2>&1 </dev/null
tail -5 $O/run.log
ls -la $O $O/* | head -30
