#!/bin/bash
# instruction mix of the embed kernel for several library variants: bash tools/exp/pmc_variants.sh shape variant...
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmcv; rm -rf $O; mkdir -p $O
S=$1; shift
cd /tmp
for v in "$@"; do
  if [ "$v" != default ]; then export SGPR_HIP_LIB=$R/variants/libsgpr_$v.so; else unset SGPR_HIP_LIB; fi
  timeout 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVE_CYCLES --output-format csv -d $O -o $v -- python $R/tools/run_embed.py $S 3 > $O/$v.log 2>&1 </dev/null
  ( cd $R; echo -n "$v: "; python tools/pmc_summary.py $O $v | grep embed_kernel | sed 's/.*{/{/' )
  f=$(find $O -name "${v}_kernel_trace.csv" | head -1)
  python3 - "$f" <<'PY'
import csv,sys
d=[(float(r['End_Timestamp'])-float(r['Start_Timestamp']))/1e3 for r in csv.DictReader(open(sys.argv[1])) if 'embed_kernel' in r['Kernel_Name']]
print('   embed us', [round(x,1) for x in d])
PY
done
