export TMPDIR=/tmp
O=gpurun_out/k5; rm -rf $O; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sequence_set or two_ranks" ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python bench.py --workload kitti5seq --steps 50 --no-cpu-baseline --no-end-to-end > $O/b_batched.json 2> $O/b_batched.err
timeout 300 python bench.py --workload kitti5seq --steps 50 --no-cpu-baseline --no-end-to-end --per-sequence-embed > $O/b_perseq.json 2> $O/b_perseq.err
SGPR_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --workload kitti5seq --steps 20 --no-cpu-baseline > $O/b_gloo2.json 2> $O/b_gloo2.err
python - <<'PY'
import json
for n in ("b_batched","b_perseq","b_gloo2"):
    try:
        r=json.loads(open('gpurun_out/k5/%s.json'%n).read().strip().splitlines()[-1])
        print(n, "ms/step %.4f"%r["ms_per_step"], "embed launch %.4f x %d"%(r["roofline"]["launch_ms"], r["roofline"]["launches_per_step"]), "tail", (r.get("roofline_tail") or {}).get("launch_ms"), r["n_gpus"])
    except Exception as e:
        print(n, "failed", e); print(open('gpurun_out/k5/%s.err'%n).read()[-1500:])
PY
