#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6f; rm -rf $O; mkdir -p $O
cd $R
cat > /tmp/plain.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from sg_pr_amd import engine, synth
sd = torch.load(os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests", "golden", "model.pth"), map_location="cpu")
eng = engine.Engine(sd)
c, l, _, _ = synth.kitti_like_sequence(4541, 100, 0)
dc, dl = torch.from_numpy(c).cuda(), torch.from_numpy(l).cuda()
for _ in range(20):
    p = eng.embed(dc, dl, 10)[0]
torch.cuda.synchronize()
PY
( cd /tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python /tmp/plain.py > $O/run.log 2>&1 </dev/null )
python tools/kstats.py $(find $O/kt -name kt_kernel_stats.csv | head -1) | head -6
python - <<'PY'
import csv, glob, os
f = glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/r6f/kt/**/kt_kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[-9]["Start_Timestamp"])
for r in rows[-9:]:
    print("%-50s start %8.1f us  dur %7.1f us  grid %s wg %s lds %s" % (r["Kernel_Name"][:50], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size"), r.get("Workgroup_Size"), r.get("LDS_Block_Size")))
PY
