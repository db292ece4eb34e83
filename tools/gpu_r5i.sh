#!/bin/bash
# round 5, pass i: sgpr_f1_max with the positives counted by bin in pass A - gates, fuzz, phases
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5i; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "f1 or pr_roc or consumers or metrics or full_sequence" > $O/f1_tests.log 2>&1
tail -8 $O/f1_tests.log
timeout 300 python tools/exp/fuzz_f1_one_call.py 60 > $O/fuzz.log 2>&1; tail -3 $O/fuzz.log
timeout 200 python tools/f1_phases.py kitti > $O/f1_phases_kitti.log 2>&1; tail -4 $O/f1_phases_kitti.log
timeout 200 python tools/f1_phases.py world > $O/f1_phases_world.log 2>&1; tail -4 $O/f1_phases_world.log
