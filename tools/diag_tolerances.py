#!/usr/bin/env python3
"""Print the OBSERVED deviations behind every tolerance gate of tests/test_gpu_parity.py (run on the GPU box), so that
the gates can be set just above what is observed."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import sgpr_oracle as oracle          # noqa: E402
from sg_pr_amd import engine, synth, metrics, allpairs   # noqa: E402

G = os.path.join(REPO, "tests", "golden")
sd = oracle.load_checkpoint(os.path.join(G, "model.pth"))
eng = engine.Engine(sd, device=0)


def packed(feats):
    centers = np.ascontiguousarray(feats[:, :3, :].transpose(0, 2, 1))
    onehot = feats[:, 3:, :]
    return centers, np.where(onehot.sum(1) > 0, onehot.argmax(1), -1).astype(np.int32)


g = np.load(os.path.join(G, "kitti3_n100_k10.npz"))
c, l = packed(g["features"])
pooled, att, emb, layers, knn = eng.embed(c, l, 10, debug=True)
layers, knn = layers.cpu().numpy(), knn.cpu().numpy()
f = g["features"]
inputs = [f[:, :3, :], g["xyz1"], g["xyz2"], f[:, 3:, :], g["sem1"], g["sem2"]]
for li, name in enumerate(["xyz1", "xyz2", "xyz3", "sem1", "sem2", "sem3"]):
    ref = g[name].transpose(0, 2, 1)
    got = layers[:, li, :, : ref.shape[2]]
    x = inputs[li].transpose(0, 2, 1)
    ok = []
    for b in range(3):
        canon = np.array([np.flatnonzero((x[b] == x[b, j]).all(-1))[0] for j in range(x.shape[1])])
        ok.append((np.sort(canon[knn[b, li]], -1) == np.sort(canon[g["knn_idx"][b, li].astype(np.int64)], -1)).all(-1))
    print("kitti3 %-5s max|d| %.3e   neighbour-set agreement %.6f" % (name, np.abs(got - ref).max(), np.mean(ok)))
print("kitti3 emb    max|d| %.3e" % np.abs(emb.cpu().numpy() - g["emb"]).max())
print("kitti3 att    max|d| %.3e" % np.abs(att.cpu().numpy() - g["att"]).max())
dp = np.abs(pooled.cpu().numpy() - g["pooled"])
print("kitti3 pooled max|d| %.3e  max rel %.3e  (|pooled| max %.3f)" % (dp.max(), (dp / (np.abs(g["pooled"]) + 1e-6)).max(), np.abs(g["pooled"]).max()))
i1 = torch.tensor(g["pair_ij"][:, 0].astype(np.int32))
i2 = torch.tensor(g["pair_ij"][:, 1].astype(np.int32))
s = eng.score_pairs(pooled, pooled, i1, i2).cpu().numpy()
m = eng.score_all_pairs(pooled, pooled).cpu().numpy().reshape(-1)
print("kitti3 scores pair-list max|d| %.3e  all-pairs max|d| %.3e  list-vs-matrix %.3e" % (
    np.abs(s - g["scores"]).max(), np.abs(m - g["scores"]).max(), np.abs(m - s).max()))
for fname in ["synth_n64_k10.npz", "synth_n100_k10.npz", "synth_n256_k20.npz"]:
    gg = np.load(os.path.join(G, fname))
    k = int(gg["k"])
    p, a, _ = eng.embed(gg["centers"], gg["labels"], k, want_att=True)
    dp = np.abs(p.cpu().numpy() - gg["pooled"])
    sc = eng.score_pairs(p[0::2].contiguous(), p[1::2].contiguous()).cpu().numpy()
    print("%-20s att %.3e pooled abs %.3e rel %.3e (max |pooled| %.2f) scores %.3e" % (
        fname, np.abs(a.cpu().numpy() - gg["att"]).max(), dp.max(), (dp / (np.abs(gg["pooled"]) + 1e-6)).max(),
        np.abs(gg["pooled"]).max(), np.abs(sc - gg["scores"]).max()))
# F1 gate of test_all_pairs_matrix_vs_oracle_and_f1
centers, labels, _, poses = synth.kitti_like_sequence(num_graphs=96, node_num=100, seed=4)
pooled, _, _ = eng.embed(centers, labels, 10)
mm = eng.score_all_pairs(pooled, pooled).cpu()
rp, _, _ = oracle.embed(sd, torch.from_numpy(synth.dense_features(centers, labels)), 10)
rm = oracle.score_all_pairs(sd, rp, rp)
gt, valid = allpairs.ground_truth_mask(allpairs.pose_distance_matrix(poses), 3)
f_hip = metrics.f1_max(gt[valid].numpy(), mm[valid].numpy())
f_ref = oracle.f1_max(gt[valid].numpy(), rm[valid].numpy())
print("M=96 all-pairs max|dscore| %.3e   F1 hip %.12f ref %.12f |d| %.3e   pooled range %.3f" % (
    (mm - rm).abs().max().item(), f_hip, f_ref, abs(f_hip - f_ref), pooled.abs().max().item()))
print("pooled magnitude over the KITTI-like set: max |e| %.3f, mean |e| %.3f" % (pooled.abs().max().item(), pooled.abs().mean().item()))
