#!/bin/bash
# round 4, pass d: the radix-histogram F1-max - tests, timings, kernel stats
set -x
mkdir -p gpurun_out/r4d
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "f1_max or device_f1 or device_roc or sequence_evaluation" > gpurun_out/r4d/pytest.log 2>&1
tail -25 gpurun_out/r4d/pytest.log
timeout 300 python tools/run_f1.py 20 check > gpurun_out/r4d/consumers.txt 2>&1; cat gpurun_out/r4d/consumers.txt
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r4d/prof -o kt -- python /root/repo/tools/run_f1.py 3 > /root/repo/gpurun_out/r4d/prof.log 2>&1)
python tools/kstats.py gpurun_out/r4d/prof/kt_kernel_stats.csv | head -24
