#!/bin/bash
# round 6, pass l: A/B of the gather's lane-group mapping (variant ggrp) on kitti00 + stress; the wide tail at 2 / 3 / 4 workgroups per CU
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6l; rm -rf $O; mkdir -p $O
cd $R
SGPR_HIP_LIB=$R/variants/libsgpr_ggrp.so timeout 900 python -m pytest tests -m gpu -x -q -k "label_lookup or shipped_graphs or synthetic_golden or config5 or lean_plans or node_cap" > $O/pytest_ggrp.log 2>&1; tail -2 $O/pytest_ggrp.log
for shape in kitti00 stress; do
for v in default ggrp default ggrp; do
  if [ "$v" != default ]; then export SGPR_HIP_LIB=$R/variants/libsgpr_$v.so; else unset SGPR_HIP_LIB; fi
  ( cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_${v}_$shape -o kt -- python $R/tools/run_embed.py $shape 30 > $O/run_${v}.log 2>&1 </dev/null )
  echo "== $v $shape"; python tools/kstats.py $(find $O/kt_${v}_$shape -name kt_kernel_stats.csv | head -1) | grep "embed" | head -1
done; done
cat > /tmp/wt.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from sg_pr_amd import engine
sd = torch.load(os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests", "golden", "model.pth"), map_location="cpu")
eng = engine.Engine(sd)
p = torch.randn(4541, 32, device="cuda")
out = torch.empty(4541, 4541, device="cuda")
eng.set_skip_mask(8192)
for _ in range(20):
    eng.score_all_pairs(p, p, out=out)
torch.cuda.synchronize()
PY
for v in default apw4 apw2; do
  if [ "$v" != default ]; then export SGPR_HIP_LIB=$R/variants/libsgpr_$v.so; else unset SGPR_HIP_LIB; fi
  ( cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktw_${v} -o kt -- python /tmp/wt.py > $O/runw_${v}.log 2>&1 </dev/null )
  echo "== wide tail $v"; python tools/kstats.py $(find $O/ktw_${v} -name kt_kernel_stats.csv | head -1) | grep "wide" | head -2
done
