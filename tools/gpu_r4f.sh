#!/bin/bash
set -x
mkdir -p gpurun_out/r4f
cd /root/repo
timeout 300 python tools/f1_phases.py world > gpurun_out/r4f/phases_world.txt 2>&1; cat gpurun_out/r4f/phases_world.txt
export TMPDIR=/tmp
O=/root/repo/gpurun_out/r4f
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt -- python /root/repo/tools/f1_phases.py world 10 > $O/kt.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM --output-format csv -d $O -o sq1 -- python /root/repo/tools/f1_phases.py world 3 > $O/sq1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM --output-format csv -d $O -o sq2 -- python /root/repo/tools/f1_phases.py world 3 > $O/sq2.log 2>&1
cd /root/repo
python tools/kstats.py gpurun_out/r4f/kt_kernel_stats.csv 2>/dev/null | grep -i "f1_\|slab" || python tools/kstats.py $(find gpurun_out/r4f -name "kt_kernel_stats.csv" | head -1) | grep -i "f1_\|slab"
python tools/pmc_summary.py gpurun_out/r4f sq1 sq2 | grep -i "f1_\|slab"
