#!/usr/bin/env python3
"""Print per-layer max|diff| / neighbour-list sanity of the embed kernel vs the golden vectors (GPU box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from sg_pr_amd import engine  # noqa: E402

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
sd = torch.load(os.path.join(root, "model.pth"), map_location="cpu")
eng = engine.Engine(sd)
g = np.load(os.path.join(root, "kitti3_n100_k10.npz"))
f = g["features"]
centers = np.ascontiguousarray(f[:, :3, :].transpose(0, 2, 1))
oh = f[:, 3:, :]
labels = np.where(oh.sum(1) > 0, oh.argmax(1), -1).astype(np.int32)
pooled, att, emb, layers, knn = eng.embed(centers, labels, 10, debug=True)
torch.cuda.synchronize()
layers, knn = layers.cpu().numpy(), knn.cpu().numpy()
names = ["xyz1", "xyz2", "xyz3", "sem1", "sem2", "sem3"]
for li, name in enumerate(names):
    ref = g[name].transpose(0, 2, 1)
    got = layers[:, li, :, : ref.shape[2]]
    bad = ~np.isfinite(got)
    diff = np.abs(np.where(bad, 0, got) - ref)
    kk = knn[:, li]
    dup = sum(len(set(r)) != len(r) for b in kk for r in b)
    print(name, "maxdiff %.3g nonfinite %d rows_bad %d | knn min %d max %d unwritten %d dup_rows %d" % (
        diff.max(), bad.sum(), (diff.max(-1) > 1e-3).sum(), kk.min(), kk.max(), (kk < 0).sum(), dup))
    if li == 0:
        rows = np.argwhere(diff.max(-1) > 1e-3)[:5]
        for b, i in rows:
            print("   row", b, i, "mine", sorted(kk[b, i]), "ref", sorted(g["knn_idx"][b, 0, i]))
print("pooled diff", np.abs(pooled.cpu().numpy() - g["pooled"]).max())
