#!/bin/bash
# Regenerate the judged artefacts under profiles/ on the GPU box (run through gpurun; outputs land in gpurun_out/prof,
# tools/collect_profiles.py <tag> then writes the summaries):  bash tools/refresh_profiles.sh
#   kitti00 : bench line, kernel stats of the same command, HBM counters (separate --pmc passes), SQ counters
#   stress / pairs128 (BASELINE configs 5 / 2): bench line, kernel stats, HBM + LDS counters
#   kitti5seq (config 4) and a 2-rank gloo run of the N > 1 path on one GPU: bench lines
#   the matrix consumers (device F1-max / ROC area): wall times and kernel stats
#   pairlist (the reference's evaluation lists): bench line + kernel stats; sgpr_f1_max per call + plan-kernel phases
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof; rm -rf $O; mkdir -p $O
cd /tmp
# (1) the counter passes, (2) their summary for exactly these sources (tools/collect_profiles.py --pmc-only writes
# profiles/pmc_hbm_latest.json with the hash of csrc/), (3) the bench lines - which report roofline.traffic / .issue only
# when that hash is the hash of the tree they run from: a refresh whose bench line lacks them fails
for shape in kitti00 stress pairs128; do
  timeout 100 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O -o fetch_$shape -- python $R/tools/run_embed.py $shape 3 > $O/fetch_$shape.log 2>&1 </dev/null
  timeout 100 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O -o write_$shape -- python $R/tools/run_embed.py $shape 3 > $O/write_$shape.log 2>&1 </dev/null
  timeout 100 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM --output-format csv -d $O -o sq1_$shape -- python $R/tools/run_embed.py $shape 3 > $O/sq1_$shape.log 2>&1 </dev/null
  timeout 100 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O -o sq3_$shape -- python $R/tools/run_embed.py $shape 3 > $O/sq3_$shape.log 2>&1 </dev/null
  timeout 100 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O -o sq2_$shape -- python $R/tools/run_embed.py $shape 3 > $O/sq2_$shape.log 2>&1 </dev/null
done
( cd $R && python tools/collect_profiles.py ${1:-rXX} --pmc-only ) > $O/collect_pmc.log 2>&1
timeout 300 python $R/bench.py > $O/bench.json 2> $O/bench.err </dev/null
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt -- python $R/bench.py --no-cpu-baseline --no-end-to-end > $O/bench_under_rocprof.json 2> $O/kt.err </dev/null
for w in stress pairs128; do
  timeout 200 python $R/bench.py --workload $w --no-cpu-baseline --steps 100 > $O/bench_$w.json 2> $O/bench_$w.err </dev/null
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt_$w -- python $R/bench.py --workload $w --no-cpu-baseline --steps 50 > $O/kt_$w.log 2>&1 </dev/null
done
timeout 300 python $R/bench.py --workload kitti5seq --no-cpu-baseline --steps 50 > $O/bench_kitti5seq.json 2> $O/bench_kitti5seq.err </dev/null
# the reference's own loop shape: its evaluation pair lists (index-pair fixture), embed once + grouped tail, with a CPU baseline
timeout 300 python $R/bench.py --workload pairlist --steps 100 > $O/bench_pairlist.json 2> $O/bench_pairlist.err </dev/null
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt_pairlist -- python $R/bench.py --workload pairlist --no-cpu-baseline --steps 50 > $O/kt_pairlist.log 2>&1 </dev/null
SGPR_BENCH_BACKEND=gloo timeout 300 python $R/bench.py --gpus 2 --steps 50 --no-cpu-baseline > $O/bench_gloo2.json 2> $O/bench_gloo2.err </dev/null
SGPR_BENCH_BACKEND=gloo timeout 300 python $R/bench.py --gpus 2 --workload pairlist --steps 50 --no-cpu-baseline > $O/bench_gloo2_pairlist.json 2> $O/bench_gloo2_pairlist.err </dev/null
python - <<PY || { echo "REFRESH FAILED: bench line without roofline.traffic (PMC profile and sources differ)"; exit 1; }
import json
r = json.loads([l for l in open("$O/bench.json") if l.startswith("{")][-1])
assert r["roofline"]["traffic"] is not None and r["roofline"].get("issue"), r["roofline"]
print("bench line carries traffic", r["roofline"]["traffic"], "and issue", r["roofline"]["issue"]["frac"])
PY
# the embed launch at the reference's operand width (wide-range instance forced, debug bit 13): kernel stats beside the
# bench line's roofline.wide_range
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt_wide -- python $R/tools/run_embed.py kitti00 5 8192 > $O/kt_wide.log 2>&1 </dev/null
# whole-sequence parity census with tie proofs (the gated form is the GPU test of the same name)
( cd $R && SGPR_SEQ_PARITY_OUT=$O/seq_parity_world.txt timeout 900 python -m pytest tests -m gpu -q -x -k "full_sequence_parity" > $O/seq_parity_pytest.log 2>&1 )
timeout 900 python $R/tools/seq_parity.py 4541 $O/seq_parity_both.txt > $O/seq_parity.log 2>&1 </dev/null
timeout 60 $R/tools/probes/tailmix_probe > $O/tailmix_probe.txt 2>&1
timeout 300 python $R/tools/run_anyshape.py $O/any_shape.txt > $O/any_shape.log 2>&1 </dev/null
# plain sgpr_embed (no node_cap promise: lean launch + hand-over) against the capped / ordered launches, five data shapes
timeout 300 python $R/tools/run_auto.py 50 > $O/plain_embed.txt 2>&1 </dev/null
# SURVEY.md 8d's calibration sample of the CPU baseline: 100 000 pairs through the oracle (minutes of CPU time; the driver's line keeps the bounded sample)
timeout 1500 python $R/bench.py --cpu-pairs 100000 --steps 20 --no-end-to-end --no-wide-range > $O/bench_cpu100k.json 2> $O/bench_cpu100k.err </dev/null
timeout 200 python $R/tools/run_f1.py 10 check > $O/consumers.log 2>&1 </dev/null
timeout 200 python $R/tools/f1_phases.py kitti > $O/f1_phases_kitti.log 2>&1 </dev/null
timeout 200 python $R/tools/f1_phases.py world > $O/f1_phases_world.log 2>&1 </dev/null
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o kt_consumers -- python $R/tools/run_f1.py 3 > $O/kt_consumers.log 2>&1 </dev/null
timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $O -o sq_consumers -- python $R/tools/run_f1.py 1 > $O/sq_consumers.log 2>&1 </dev/null
for c in FETCH_SIZE WRITE_SIZE; do timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O -o ${c}_consumers -- python $R/tools/run_f1.py 1 > $O/${c}_consumers.log 2>&1 </dev/null; done
# fuzzers on the round's new paths: sgpr_f1_max (3 x 120 cases, a third in pose mode) and architectures inside sgpr_wide.hip's limits
( for sd in 41 42 43; do timeout 400 python $R/tools/exp/fuzz_f1_one_call.py $sd 120 2>&1 | grep -v amdgpu.ids | tail -1; done ) > $O/fuzz_f1.txt 2>&1 </dev/null
( cd $R && timeout 900 python tools/exp/fuzz_anyshape.py 40 wide 2>&1 | grep -v amdgpu.ids ) > $O/fuzz_anyshape_wide.txt 2>&1 </dev/null
# per-wave / per-workgroup time stamps, when the variant libraries travelled along (tools/build_variant.sh stamps
# -DSGPR_F1_SCAN_STAMPS=1; ... estamps -DSGPR_EMBED_STAMPS=1)
if [ -f $R/variants/libsgpr_stamps.so ]; then
  for k in kitti world; do echo "== $k"; SGPR_HIP_LIB=$R/variants/libsgpr_stamps.so timeout 200 python $R/tools/exp/f1_scan_timeline.py $k 2>&1 | grep -v amdgpu.ids; done > $O/f1_scan_timeline.txt
fi
if [ -f $R/variants/libsgpr_estamps.so ]; then
  SGPR_HIP_LIB=$R/variants/libsgpr_estamps.so timeout 200 python $R/tools/exp/embed_timeline.py 2>&1 | grep -v amdgpu.ids > $O/embed_timeline.txt
fi
ls $O | head -80
