#!/usr/bin/env python3
"""Time the consumers of the KITTI-00-sized score matrix: device F1-max / top-k vs the host path (GPU box only)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from sg_pr_amd import allpairs, engine, metrics, synth  # noqa: E402

sd = torch.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "model.pth"), map_location="cpu")
eng = engine.Engine(sd)
c, l, _, poses = synth.kitti_like_sequence(4541, 100, 0)
order, cap = eng.size_order(c, l, 10)
pooled = eng.embed(c, l, 10, node_cap=cap, order=order)[0]
m = eng.score_all_pairs(pooled, pooled)
xz = allpairs.pose_xz(poses).cuda()
torch.cuda.synchronize()
for _ in range(2):
    f_dev, passes = metrics.f1_max_device(eng, m, pose_xz=xz)
torch.cuda.synchronize()
t0 = time.perf_counter()
f_dev, passes = metrics.f1_max_device(eng, m, pose_xz=xz)
torch.cuda.synchronize()
t_dev = time.perf_counter() - t0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
eng.pair_histogram(m, pose_xz=xz)
e1.record()
torch.cuda.synchronize()
t0 = time.perf_counter()
mh = m.cpu().numpy()
t_copy = time.perf_counter() - t0
t0 = time.perf_counter()
d = allpairs.pose_distance_matrix(poses)
gt, valid = allpairs.ground_truth_mask(d, 3)
f_host = metrics.f1_max(gt[valid].numpy(), mh[valid.numpy()])
t_host = time.perf_counter() - t0
print("F1-max device %.12f (%d passes, %.3f ms wall, one pass incl. D2H of the counts %.3f ms)  host %.12f (D2H %.1f ms + sort %.0f ms)"
      % (f_dev, passes, t_dev * 1e3, e0.elapsed_time(e1), f_host, t_copy * 1e3, t_host * 1e3))
for k in (1, 8):
    eng.topk_rows(m, k=k, window=50)
    e0.record()
    for _ in range(10):
        eng.topk_rows(m, k=k, window=50)
    e1.record()
    torch.cuda.synchronize()
    print("top-%d loop-closure candidates of 4541 rows: %.3f ms" % (k, e0.elapsed_time(e1) / 10))
