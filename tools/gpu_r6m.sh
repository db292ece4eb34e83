#!/bin/bash
# round 6, pass m: the wide tail with one / two row graphs interleaved in program order (variant apwni1 = one)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6m; rm -rf $O; mkdir -p $O
cd $R
( timeout 600 python -m pytest tests -m gpu -x -q -k "f16_range_guard or shipped_graphs or larger_architectures" ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
cat > /tmp/wt.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from sg_pr_amd import engine, synth
sd = torch.load(os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests", "golden", "model.pth"), map_location="cpu")
eng = engine.Engine(sd)
c, l, _, _ = synth.kitti_like_sequence(4541, 100, 0)
p = eng.embed(c, l, 10)[0]
out = torch.empty(4541, 4541, device="cuda")
eng.set_skip_mask(8192)
for _ in range(30):
    eng.score_all_pairs(p, p, out=out)
torch.cuda.synchronize()
PY
for v in default apwni1 default apwni1; do
  if [ "$v" != default ]; then export SGPR_HIP_LIB=$R/variants/libsgpr_$v.so; else unset SGPR_HIP_LIB; fi
  ( cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktw_${v} -o kt -- python /tmp/wt.py > $O/runw_${v}.log 2>&1 </dev/null )
  echo "== wide tail $v"; python tools/kstats.py $(find $O/ktw_${v} -name kt_kernel_stats.csv | head -1) | grep "wide" | head -1
done
