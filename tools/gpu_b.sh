#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/b; rm -rf $O; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1
tail -15 $O/pytest.log
timeout 300 python tools/diag_tolerances.py > $O/diag.log 2>&1; grep -E "scores|M=96" $O/diag.log
timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-end-to-end > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
r=json.loads(open('gpurun_out/b/bench.json').read().strip().splitlines()[-1])
print("ms/step", r["ms_per_step"], "embed", r["roofline"]["launch_ms"], "tail", r["roofline_tail"]["launch_ms"], "tail GB/s", r["roofline_tail"]["achieved"])
PY
cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --no-cpu-baseline --no-end-to-end > $GRAFT_REPO_ROOT/$O/kt.log 2>&1
cd $GRAFT_REPO_ROOT; python - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/b/**/kt_kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        print("%-70s calls %5s avg %9.1f us  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
