#!/bin/bash
# round 6, pass k: the three-plane (24-bit operand) dense tail: tests + bench line with roofline.wide_range
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6k; rm -rf $O; mkdir -p $O
cd $R
( timeout 900 python -m pytest tests -m gpu -x -q -s -k "f16_range_guard or wide_range or all_pairs_matrix or shipped_graphs" ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log; grep -E "wide tail|tail vs float64|f16 planes vs wide" $O/pytest.log | cut -c1-200
timeout 400 python bench.py --no-cpu-baseline --no-end-to-end --steps 100 > $O/bench.json 2> $O/bench.err </dev/null; tail -c 400 $O/bench.err
python - <<PY
import json
r = json.loads([l for l in open("$O/bench.json") if l.startswith("{")][-1])
print("kitti00: step %.4f ms value %.3e  embed %.4f tail %.4f" % (r["ms_per_step"], r["value"], r["kernel_durations"]["embed_call_ms"], r["kernel_durations"]["tail_call_ms"]))
print(json.dumps(r["roofline"]["wide_range"], indent=1))
PY
