#!/bin/bash
# round 6, pass h: full GPU suite on the round's sources, bench lines (kitti00 default incl. cpu baseline, stress, pairlist,
# 2-rank gloo pairlist), the empty second pass against its grid (variant redo8)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6h; rm -rf $O; mkdir -p $O
cd $R
( SGPR_SEQ_PARITY_OUT=$O/seq_parity_world.txt timeout 2400 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
grep -E "config 4 full size|config 5 full size" $O/pytest.log | cut -c1-300
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err </dev/null; tail -c 600 $O/bench.err
python - <<PY
import json
r = json.loads([l for l in open("$O/bench.json") if l.startswith("{")][-1])
print("kitti00: step %.4f ms value %.3e  embed %.4f tail %.4f" % (r["ms_per_step"], r["value"], r["kernel_durations"]["embed_call_ms"], r["kernel_durations"]["tail_call_ms"]))
print(" device_prep_ms_once", r["config"].get("device_prep_ms_once"), " embed_calls_ms", r["config"].get("embed_calls_ms"))
print(" end_to_end", {k: (round(v["ms_per_step"], 4) if isinstance(v, dict) else v) for k, v in (r.get("end_to_end") or {}).items() if k != "note"})
print(" cpu", r.get("cpu_baseline", {}).get("value"), r.get("cpu_baseline", {}).get("cores"))
PY
for w in stress pairlist; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline --steps 100 > $O/bench_$w.json 2> $O/bench_$w.err </dev/null
  python -c "
import json
r = json.loads([l for l in open('$O/bench_$w.json') if l.startswith('{')][-1])
print('$w: step %.4f ms value %.3e embed %.4f' % (r['ms_per_step'], r['value'], r['kernel_durations']['embed_call_ms']), r['config'].get('device_prep_ms_once'))"
done
SGPR_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --workload pairlist --steps 50 --no-cpu-baseline > $O/bench_gloo2_pairlist.json 2> $O/bench_gloo2_pairlist.err </dev/null
python -c "
import json
r = json.loads([l for l in open('$O/bench_gloo2_pairlist.json') if l.startswith('{')][-1])
print('gloo2 pairlist: step %.4f ms value %.3e' % (r['ms_per_step'], r['value']), r.get('pairlist'))" 2>&1 | cut -c1-400
for v in default redo8; do
  if [ "$v" != default ]; then export SGPR_HIP_LIB=$R/variants/libsgpr_$v.so; else unset SGPR_HIP_LIB; fi
  ( cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_${v} -o kt -- python $R/tools/run_embed.py kitti00 30 > $O/run_${v}.log 2>&1 </dev/null )
  echo "== $v"; python tools/kstats.py $(find $O/kt_${v} -name kt_kernel_stats.csv | head -1) | grep "embed" | head -3
done
