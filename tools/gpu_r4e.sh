#!/bin/bash
set -x
mkdir -p gpurun_out/r4e
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "f1_max or device_f1 or device_roc or sequence_evaluation" > gpurun_out/r4e/pytest.log 2>&1
tail -5 gpurun_out/r4e/pytest.log
timeout 300 python tools/f1_phases.py kitti > gpurun_out/r4e/phases_kitti.txt 2>&1; cat gpurun_out/r4e/phases_kitti.txt
timeout 300 python tools/f1_phases.py world > gpurun_out/r4e/phases_world.txt 2>&1; cat gpurun_out/r4e/phases_world.txt
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r4e/prof -o kt -- python /root/repo/tools/f1_phases.py world 10 > /root/repo/gpurun_out/r4e/prof.log 2>&1)
python tools/kstats.py gpurun_out/r4e/prof/kt_kernel_stats.csv | grep -i "f1_\|slab" 
