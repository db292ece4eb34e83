#!/bin/bash
# instruction mix + instruction-cache counters of the embed kernel: bash tools/exp/pmc_quick.sh [shape] [variant]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmcq; rm -rf $O; mkdir -p $O
S=${1:-kitti00}
if [ -n "$2" ]; then export SGPR_HIP_LIB=$R/variants/libsgpr_$2.so; fi
cd /tmp
timeout 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O -o a -- python $R/tools/run_embed.py $S 3 > $O/a.log 2>&1 </dev/null
timeout 120 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH --output-format csv -d $O -o b -- python $R/tools/run_embed.py $S 3 > $O/b.log 2>&1 </dev/null
timeout 120 rocprofv3 --kernel-trace --pmc SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SENDMSG --output-format csv -d $O -o c -- python $R/tools/run_embed.py $S 3 > $O/c.log 2>&1 </dev/null
cd $R
python tools/pmc_summary.py $O a b c | grep embed_kernel
tail -2 $O/b.log $O/c.log | grep -i -E "error|invalid|not" | head
