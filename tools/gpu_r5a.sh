#!/bin/bash
# round 5, first GPU pass (one gpurun call): the GPU test suite on the new sources, same-box A/B of the round's variants
# (kitti00 bench under rocprofv3 --kernel-trace --stats), the stress / pairs128 shapes, counters of the tail variants,
# the co-execution probes.  Output: gpurun_out/r5a/
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5a; rm -rf $O; mkdir -p $O
cd $R
( SGPR_SEQ_PARITY_OUT=$O/seq_parity_world.txt timeout 1700 python -m pytest tests -m gpu -x -q -s ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
grep -E "config 5 full size|pair list 02|proven ties|random shapes" $O/pytest.log | cut -c1-400

bench() {   # name, extra bench flags...; SGPR_HIP_LIB selects the library
  local name=$1; shift
  ( cd /tmp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$name -o kt -- python $R/bench.py --steps ${STEPS:-60} --warmup 5 --no-cpu-baseline --no-end-to-end "$@" > $O/bench_$name.json 2> $O/bench_$name.err </dev/null )
  echo "== $name: $(python -c "import json,sys; r=json.loads([l for l in open('$O/bench_$name.json') if l.startswith('{')][-1]); print('step %.4f ms  embed %.4f  tail %s' % (r['ms_per_step'], r['kernel_durations']['embed_call_ms'], r['kernel_durations']['tail_call_ms']))" 2>&1 | tail -1)"
  python tools/kstats.py $(find $O/kt_$name -name kt_kernel_stats.csv | head -1) | head -${HEAD:-4}
}
unset SGPR_HIP_LIB
bench default
bench default_nofuse --no-fused-prep
for v in nostage dbl8 ni2 ch2 ni2occ3 ni2ch2occ3 occ3; do
  if [ -f $R/variants/libsgpr_$v.so ]; then SGPR_HIP_LIB=$R/variants/libsgpr_$v.so bench $v; fi
done
unset SGPR_HIP_LIB
bench default_again
STEPS=100 bench stress --workload stress
STEPS=200 bench pairs128 --workload pairs128
STEPS=40 bench kitti5seq --workload kitti5seq
STEPS=100 bench pairlist --workload pairlist

# counters: tail (default vs the interleaved two-chain variant), stress traffic
pmc() {  # name shape counters...
  local name=$1 shape=$2; shift 2
  ( cd /tmp; timeout 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O -o $name -- python $R/tools/run_embed.py $shape 3 > $O/$name.log 2>&1 </dev/null )
}
pmc sq2_default kitti00 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA
if [ -f $R/variants/libsgpr_ni2ch2occ3.so ]; then SGPR_HIP_LIB=$R/variants/libsgpr_ni2ch2occ3.so pmc sq2_ni2ch2occ3 kitti00 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA; fi
if [ -f $R/variants/libsgpr_ni2.so ]; then SGPR_HIP_LIB=$R/variants/libsgpr_ni2.so pmc sq2_ni2 kitti00 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA; fi
unset SGPR_HIP_LIB
pmc fetch_stress stress FETCH_SIZE
pmc write_stress stress WRITE_SIZE
pmc sq2_stress stress SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA
python tools/pmc_summary.py $O sq2_default sq2_ni2 sq2_ni2ch2occ3 fetch_stress write_stress sq2_stress 2>&1 | tee $O/pmc_summary.txt | cut -c1-330

for p in tailmix_probe overlap_probe coexec_probe; do
  echo "== $p"; timeout 60 $R/tools/probes/$p 2>&1 | tee $O/$p.txt | tail -12
done
