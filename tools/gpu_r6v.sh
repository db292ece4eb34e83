#!/bin/bash
# round 6, pass v: the matrix-core any-shape embed at {12, 128/128/64, 32/32}: per-graph time at 1024 / 4096 graphs, SQ counters
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6v; rm -rf $O; mkdir -p $O; cd $R
for g in 1024 4096; do python tools/exp/wide_embed_run.py $g 2>&1 | grep -v amdgpu; done
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  tag=$(echo $set | cut -d' ' -f1)
  ( cd /tmp; timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$tag -o p -- python $R/tools/exp/wide_embed_run.py 1024 > $O/run_$tag.log 2>&1 </dev/null )
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/pmc_*/**/p_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"][:70]
        if "wide_embed" in n: acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n in acc:
    print(n[:60], " ".join("%s %.2fM" % (c.replace("SQ_", ""), sum(x) / len(x) / 1e6) for c, x in sorted(acc[n].items())))
PY
