# tests + bench + HBM counters of the all-pairs tail
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/tt; rm -rf $O; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-end-to-end > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json
r=json.loads(open('gpurun_out/tt/bench.json').read().strip().splitlines()[-1])
print("ms/step %.4f"%r["ms_per_step"], "embed %.4f"%r["roofline"]["launch_ms"], "tail", (r.get("roofline_tail") or {}).get("launch_ms"))
PY
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
timeout 100 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O -o p_$c -- python $R/tools/run_embed.py kitti00 3 > $O/p_$c.log 2>&1 </dev/null
done
cd $R
python - <<'PY'
import csv,glob,collections
for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob('gpurun_out/tt/**/p_%s_counter_collection.csv'%c, recursive=True)[0]
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:45]].append(float(r["Counter_Value"]))
    for k,v in agg.items():
        if 'score_all' in k or 'embed_kernel' in k: print(c, k, round(sum(v)/len(v),1), 'KiB per launch (summed over XCDs?) n=',len(v))
PY
