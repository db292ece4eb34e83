#!/bin/bash
# tests + the three bench workloads (outputs under gpurun_out/c/)
export TMPDIR=/tmp
O=gpurun_out/c; rm -rf $O; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log
for w in kitti00 stress pairs128; do
timeout 300 python bench.py --workload $w --steps 100 --no-cpu-baseline --no-end-to-end > $O/bench_$w.json 2> $O/bench_$w.err; python - $w <<'PY'
import json,sys
r=json.loads(open('gpurun_out/c/bench_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "ms/step %.4f"%r["ms_per_step"], "embed %.4f"%r["roofline"]["launch_ms"], "tail", (r.get("roofline_tail") or {}).get("launch_ms"))
PY
done
