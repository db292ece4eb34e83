#!/usr/bin/env python3
"""Time the device-side consumers of the KITTI-00-sized score matrix: F1-max and ROC area from threshold counts
(sg_pr_amd.metrics.f1_max_device / roc_auc_device / pr_roc_device).  Prints the wall time per call and per engine call."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from sg_pr_amd import allpairs, engine, metrics, synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n, k, g = 100, 10, 4541
sd = torch.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "model.pth"), map_location="cpu")
eng = engine.Engine(sd)
c, l, _, poses = synth.kitti_like_sequence(g, n, 0)
order, cap = eng.size_order(c, l, k)
p = eng.embed(torch.from_numpy(c).cuda(), torch.from_numpy(l).cuda(), k, node_cap=cap, order=order)[0]
mat = eng.score_all_pairs(p, p)
xz = allpairs.pose_xz(poses).cuda()
torch.cuda.synchronize()
calls = []


def timed(name):
    orig = getattr(eng, name)

    def f(*a, **kw):
        torch.cuda.synchronize()
        t = time.perf_counter()
        out = orig(*a, **kw)
        torch.cuda.synchronize()
        calls.append((name, time.perf_counter() - t))
        return out
    return orig, f


for name, fn in (("f1_max one call", lambda: metrics.f1_max_device(eng, mat, pose_xz=xz)),
                 ("f1_max multi-call", lambda: metrics.f1_max_device(eng, mat, pose_xz=xz, one_call=False)),
                 ("roc_auc_device", lambda: metrics.roc_auc_device(eng, mat, pose_xz=xz)),
                 ("pr_roc_device", lambda: metrics.pr_roc_device(eng, mat, pose_xz=xz))):
    fn()
    saved = {n: timed(n) for n in ("pair_positives", "pair_threshold_counts")}
    for n, (_, f) in saved.items():
        setattr(eng, n, f)
    calls.clear()
    out = fn()
    per_call = list(calls)
    for n, (orig, _) in saved.items():
        setattr(eng, n, orig)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("%-16s result %s  %.3f ms per call; engine calls: %s" %
          (name, out, dt * 1e3, " ".join("%s %.3f ms" % (n.replace("pair_", ""), x * 1e3) for n, x in per_call)))
if len(sys.argv) > 2:       # exact check against the sorted host path
    gt, valid = allpairs.ground_truth_mask(allpairs.pose_distance_matrix(poses), 3.0)
    print("host f1 %.15f auc %.15f" % (metrics.f1_max(gt[valid].numpy(), mat.cpu()[valid].numpy()),
                                      metrics.roc_auc(gt[valid].numpy(), mat.cpu()[valid].numpy())))
