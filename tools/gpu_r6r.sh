#!/bin/bash
# round 6, pass r: the narrow instance of the matrix-core any-shape embed at 4 / 2 waves per SIMD (variant wnocc2), tests
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6r; rm -rf $O; mkdir -p $O
cd $R
( timeout 1200 python -m pytest tests -m gpu -x -q -k "matrix_core_any_shape or larger_architectures" ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log | cut -c1-300
cat > /tmp/w13.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from sg_pr_amd import engine, synth
sd = torch.load(os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests", "golden", "model.pth"), map_location="cpu")
sd = {k[7:] if k.startswith("module.") else k: v for k, v in sd.items()}
sd13 = {k: v.clone() for k, v in sd.items()}
w = sd13["dgcnn_f_conv1.0.weight"]
sd13["dgcnn_f_conv1.0.weight"] = torch.cat((w.reshape(w.shape[0], 2, 12), torch.zeros(w.shape[0], 2, 1)), dim=2).reshape(w.shape[0], 26, 1, 1)
any13 = engine.Engine(sd13, engine.SgprDims(13, 64, 64, 32, 16, 16))
c, l, _, _ = synth.kitti_like_sequence(4541, 100, 0)
cd, ld = torch.from_numpy(c).cuda(), torch.from_numpy(l).cuda()
for _ in range(6):
    p = any13.embed(cd, ld, 10)[0]
torch.cuda.synchronize()
PY
for v in default wnocc2 default wnocc2; do
  if [ "$v" != default ]; then export SGPR_HIP_LIB=$R/variants/libsgpr_$v.so; else unset SGPR_HIP_LIB; fi
  ( cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$v -o kt -- python /tmp/w13.py > $O/run_$v.log 2>&1 </dev/null )
  echo "== $v"; python tools/kstats.py $(find $O/kt_$v -name kt_kernel_stats.csv | head -1) | head -2
done
