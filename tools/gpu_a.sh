#!/bin/bash
# round-2 first GPU pass: tests, observed tolerances, probes, bench lines (outputs under gpurun_out/a/)
export TMPDIR=/tmp
O=gpurun_out/a; rm -rf $O; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
timeout 300 python tools/diag_tolerances.py > $O/diag.log 2>&1; tail -25 $O/diag.log
timeout 60 tools/probes/f16_split_probe > $O/probe.log 2>&1; cat $O/probe.log
timeout 300 python bench.py --steps 100 > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json
SGPR_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 20 --no-cpu-baseline > $O/bench_gloo2.json 2> $O/bench_gloo2.err; tail -c 1500 $O/bench_gloo2.json; tail -3 $O/bench_gloo2.err
timeout 300 python bench.py --workload kitti5seq --steps 20 --no-cpu-baseline > $O/bench_5seq.json 2> $O/bench_5seq.err; tail -c 1500 $O/bench_5seq.json; tail -3 $O/bench_5seq.err
timeout 200 python bench.py --workload stress --steps 20 --no-cpu-baseline > $O/bench_stress.json 2> $O/bench_stress.err; tail -c 800 $O/bench_stress.json
timeout 200 python bench.py --workload pairs128 --steps 200 --no-cpu-baseline > $O/bench_pairs128.json 2> $O/bench_pairs128.err; tail -c 800 $O/bench_pairs128.json
