#!/bin/bash
# SQ counter passes (millions per launch) for the kernels of one KITTI-00 step:  bash tools/pmc.sh <out dir under gpurun_out>
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $O
cd /tmp
timeout 100 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM --output-format csv -d $O -o sq1 -- python $GRAFT_REPO_ROOT/tools/run_embed.py kitti00 3 > $O/sq1.log 2>&1 </dev/null
timeout 100 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O -o sq2 -- python $GRAFT_REPO_ROOT/tools/run_embed.py kitti00 3 > $O/sq2.log 2>&1 </dev/null
timeout 100 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O -o sq3 -- python $GRAFT_REPO_ROOT/tools/run_embed.py kitti00 3 > $O/sq3.log 2>&1 </dev/null
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $O sq1 sq2 sq3
