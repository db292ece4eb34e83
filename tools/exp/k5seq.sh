export TMPDIR=/tmp
O=gpurun_out/k5; rm -rf $O; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sequence_set or two_ranks or abi or export" ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for v in "" "--per-sequence-tails" "" "--per-sequence-tails"; do
timeout 300 python bench.py --workload kitti5seq --steps 50 --no-cpu-baseline --no-end-to-end $v > $O/b.json 2> $O/b.err
python - "$v" <<'PY'
import json,sys
try:
    r=json.loads(open('gpurun_out/k5/b.json').read().strip().splitlines()[-1])
    print(sys.argv[1] or "batched", "ms/step %.4f"%r["ms_per_step"], "embed %.4f x %d"%(r["roofline"]["launch_ms"], r["roofline"]["launches_per_step"]), "tail", (r.get("roofline_tail") or {}).get("launch_ms"), (r.get("roofline_tail") or {}).get("calls_per_step"))
except Exception as e:
    print("failed", e); print(open('gpurun_out/k5/b.err').read()[-1500:])
PY
done
timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-end-to-end > $O/b00.json 2> $O/b00.err; python -c "
import json; r=json.loads(open('gpurun_out/k5/b00.json').read().strip().splitlines()[-1]); print('kitti00 ms/step %.4f'%r['ms_per_step'], 'tail', r['roofline_tail']['launch_ms'])"
