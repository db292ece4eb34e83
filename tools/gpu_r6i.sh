#!/bin/bash
# round 6, pass i: rest of the GPU suite after the config-4 gate fix; size_order kernels' durations
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6i; rm -rf $O; mkdir -p $O
cd $R
( SGPR_SEQ_PARITY_OUT=$O/seq_parity_world.txt timeout 2400 python -m pytest tests -m gpu -x -q ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
grep -E "config 4 full size|config 5 full size" $O/pytest.log | cut -c1-300
cat > /tmp/so.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from sg_pr_amd import engine, synth
sd = torch.load(os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests", "golden", "model.pth"), map_location="cpu")
eng = engine.Engine(sd)
c, l, _, _ = synth.kitti_like_sequence(4541, 100, 0)
dc, dl = torch.from_numpy(c).cuda(), torch.from_numpy(l).cuda()
for _ in range(20):
    o = eng.size_order_device(dc, dl, None, 100, 10)
torch.cuda.synchronize()
PY
( cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python /tmp/so.py > $O/run.log 2>&1 </dev/null )
python tools/kstats.py $(find $O/kt -name kt_kernel_stats.csv | head -1) | head -4
