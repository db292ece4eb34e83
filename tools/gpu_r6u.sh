#!/bin/bash
# round 6, pass u: SQ counters of sgpr_f1_max's passes, current sources against the variant(s) named in $VARIANTS
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6u; rm -rf $O; mkdir -p $O
cd $R
for v in default ${VARIANTS:-nocull}; do
  if [ "$v" != default ]; then export SGPR_HIP_LIB=$R/variants/libsgpr_$v.so; else unset SGPR_HIP_LIB; fi
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"; do
    tag=$(echo $set | cut -d' ' -f1)
    ( cd /tmp; timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_${v}_$tag -o p -- python $R/tools/f1_phases.py kitti 3 > $O/run_${v}_$tag.log 2>&1 </dev/null )
  done
  echo "== $v"
  python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/pmc_${v}_*/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0]
        if "f1_" in n or "slab" in n:
            acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n in acc:
    print(n[:28].ljust(28), " ".join("%s %.2fM" % (c.replace("SQ_", ""), sum(x) / len(x) / 1e6) for c, x in sorted(acc[n].items())))
PY
done
