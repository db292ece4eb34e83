#!/bin/bash
# round 6, pass b: what binds embed_big_kernel on the stress shape - phase costs by running an idempotent phase twice
# (variants bigsel2 / biggemm2), and the SQ / LDS counters of the launch.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6b; rm -rf $O; mkdir -p $O
cd $R
for v in default bigsel2 biggemm2; do
  if [ "$v" != default ]; then export SGPR_HIP_LIB=$R/variants/libsgpr_$v.so; else unset SGPR_HIP_LIB; fi
  ( cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_${v} -o kt -- python $R/tools/run_embed.py stress 20 > $O/run_${v}.log 2>&1 </dev/null )
  echo "== $v stress"; python tools/kstats.py $(find $O/kt_${v} -name kt_kernel_stats.csv | head -1) | head -2
done
unset SGPR_HIP_LIB
pmc() {  # name shape counters...
  local name=$1 shape=$2; shift 2
  ( cd /tmp; timeout 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O -o $name -- python $R/tools/run_embed.py $shape 3 > $O/$name.log 2>&1 </dev/null )
}
pmc sq_a stress SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA
pmc sq_b stress SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA
pmc sq_c stress SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAVES SQ_INSTS_SMEM GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL
python tools/pmc_summary.py $O sq_a sq_b sq_c 2>&1 | tee $O/pmc_summary.txt | cut -c1-400
