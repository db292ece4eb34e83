#!/bin/bash
# round 4, pass b: the new pair-list tests, the full-size parity test, the pairlist bench with a kernel trace
set -x
mkdir -p gpurun_out/r4b
cd /root/repo
SGPR_SEQ_PARITY_OUT=gpurun_out/r4b/seq_parity.txt timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "grouped_pair_list or reference_pair_lists or full_sequence_parity" > gpurun_out/r4b/pytest.log 2>&1
tail -15 gpurun_out/r4b/pytest.log
timeout 300 python bench.py --workload pairlist --steps 50 --warmup 5 > gpurun_out/r4b/bench_pairlist.json 2> gpurun_out/r4b/bench_pairlist.err
tail -3 gpurun_out/r4b/bench_pairlist.err; cat gpurun_out/r4b/bench_pairlist.json
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r4b/bench_kitti00.json 2> gpurun_out/r4b/bench_kitti00.err
tail -3 gpurun_out/r4b/bench_kitti00.err; cat gpurun_out/r4b/bench_kitti00.json
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r4b/prof_pairlist -o pl -- python /root/repo/bench.py --workload pairlist --steps 20 --warmup 2 --no-cpu-baseline --prewarm 0.3 > /root/repo/gpurun_out/r4b/prof_pairlist.log 2>&1)
python tools/kstats.py $(find gpurun_out/r4b/prof_pairlist -name '*kernel_stats.csv' | head -1) > gpurun_out/r4b/pairlist_kernel_stats.txt 2>&1; cat gpurun_out/r4b/pairlist_kernel_stats.txt | head -30
