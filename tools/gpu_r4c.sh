#!/bin/bash
# round 4, pass c: launch-shape probe, the whole GPU suite on the new second-pass dispatcher / split-launch order, bench lines
set -x
mkdir -p gpurun_out/r4c
cd /root/repo
./tools/probes/launch_shape_probe > gpurun_out/r4c/launch_shape.txt 2>&1; cat gpurun_out/r4c/launch_shape.txt
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r4c/pytest.log 2>&1
tail -8 gpurun_out/r4c/pytest.log
for w in kitti00 pairlist pairs128; do
  timeout 300 python bench.py --workload $w --steps 100 --warmup 5 --no-cpu-baseline --no-end-to-end > gpurun_out/r4c/bench_$w.json 2> gpurun_out/r4c/bench_$w.err
  python - <<PY
import json
r=json.loads([l for l in open("gpurun_out/r4c/bench_$w.json") if l.startswith("{")][-1])
print("$w", "ms_per_step %.4f" % r["ms_per_step"], r["kernel_durations"], r.get("pairlist"))
PY
done
export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r4c/prof -o kt -- python /root/repo/bench.py --steps 50 --warmup 2 --no-cpu-baseline --no-end-to-end --prewarm 0.5 > /root/repo/gpurun_out/r4c/prof.log 2>&1)
python tools/kstats.py gpurun_out/r4c/prof/kt_kernel_stats.csv | head
