#!/bin/bash
# round 6, pass c: the serial sections of embed_big_kernel on the stress shape by compiled-in skip masks
# (results of the skip variants are garbage; only their launch time is read)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6c; rm -rf $O; mkdir -p $O
cd $R
for v in default ${VARIANTS:-skip16 skip32 skipsem skipatt skipend}; do
  if [ "$v" != default ]; then export SGPR_HIP_LIB=$R/variants/libsgpr_$v.so; else unset SGPR_HIP_LIB; fi
  ( cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_${v} -o kt -- python $R/tools/run_embed.py ${SHAPE:-stress} 20 > $O/run_${v}.log 2>&1 </dev/null )
  echo "== $v"; python tools/kstats.py $(find $O/kt_${v} -name kt_kernel_stats.csv | head -1) | grep embed | head -2
done
