#!/usr/bin/env python3
"""Fuzz of sgpr_f1_max (one call: radix histogram on the score's bit pattern + refine pass) against the sorted host
computation: random shapes incl. 1 x 1, label densities, padded leading dimensions, labels from gt bytes or from poses
(row0 offsets), and score distributions aimed at the histogram - rounded scores (mass ties), saturated 1.0 / 0.0,
scores crowded into one bin, scores on the bin edges of the key map, denormals.   python fuzz_f1_one_call.py seed trials"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from sg_pr_amd import engine, metrics  # noqa: E402

sd = torch.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "model.pth"), map_location="cpu")
eng = engine.Engine(sd)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 100
worst, n, fell_back = 0.0, 0, 0
for trial in range(trials):
    big = trial % 10 == 9
    r = int(rng.integers(1, 3000 if big else 700))
    m = int(rng.integers(1, 3000 if big else 1200))
    ld = m + int(rng.integers(0, 9))
    p_pos = float(rng.choice([0.0005, 0.01, 0.1, 0.5, 0.95]))
    mode = trial % 8
    use_pose = trial % 3 == 0
    if use_pose:
        # a random walk: label 1 below 3 m, 0 above 20 m, ignored in between (eval_batch.py:30-36 semantics)
        row0 = int(rng.integers(0, max(1, m - r + 1))) if m >= r else 0
        xz = np.cumsum(rng.normal(0, float(rng.choice([0.3, 2.0, 8.0])), (max(m, row0 + r), 2)), axis=0)
        d = np.linalg.norm(xz[row0:row0 + r, None, :] - xz[None, :m, :], axis=2)
        lab = np.where(d <= 3.0, 1, np.where(d >= 20.0, 0, -1)).astype(np.int8)
    else:
        row0 = 0
        lab = np.where(rng.random((r, m)) < 0.1, -1, (rng.random((r, m)) < p_pos).astype(np.int8)).astype(np.int8)
    z = rng.normal(0, 3, (r, m)) + float(rng.choice([0.0, 1.0, 3.0])) * (lab == 1)
    sc = (1 / (1 + np.exp(-z))).astype(np.float32)
    if mode == 1:
        sc = np.round(sc, 2).astype(np.float32)
    elif mode == 2:
        sc = np.where(rng.random((r, m)) < 0.3, np.float32(1.0), sc).astype(np.float32)
        sc = np.where(rng.random((r, m)) < 0.1, np.float32(0.0), sc).astype(np.float32)
    elif mode == 3:      # everything inside one or two bins of the key map
        base = np.float32(rng.choice([0.3, 0.75, 0.999, 1e-4]))
        sc = (base + (rng.integers(0, 40, (r, m)) * np.spacing(base))).astype(np.float32)
    elif mode == 4:      # bit patterns on the seams: low 17 bits all zero / all one
        u = sc.view(np.uint32)
        u = np.where(rng.random((r, m)) < 0.5, u & np.uint32(0xFFFE0000), u | np.uint32(0x0001FFFF))
        sc = np.minimum(u.view(np.float32), np.float32(1.0))
    elif mode == 5:      # tiny scores incl. denormals, and scores just below 1
        sc = np.where(rng.random((r, m)) < 0.5, (sc * np.float32(1e-38)).astype(np.float32),
                      np.nextafter(np.float32(1.0), np.float32(0.0)) - (sc * np.float32(1e-6))).astype(np.float32)
    elif mode == 6:
        sc = np.round(sc, 4).astype(np.float32)
    if not (lab == 1).any():
        continue
    buf = torch.zeros(r, ld, device="cuda")
    buf[:, :m] = torch.from_numpy(sc).cuda()
    keep = lab.ravel() >= 0
    want = metrics.f1_max(lab.ravel()[keep], sc.ravel()[keep])
    if use_pose:
        res = eng.f1_max(buf[:, :m], row0=row0, pose_xz=torch.from_numpy(xz).cuda(), d_pos=3.0, d_neg=20.0, gt=None)
        got, _ = metrics.f1_max_device(eng, buf[:, :m], pose_xz=torch.from_numpy(xz).cuda(), row0=row0)
    else:
        res = eng.f1_max(buf[:, :m], gt=torch.from_numpy(lab).cuda())
        got, _ = metrics.f1_max_device(eng, buf[:, :m], gt=torch.from_numpy(lab))
    status = int(res[1])
    fell_back += status == 1
    assert status in (0, 1), (trial, status)
    if status == 0:
        assert abs(float(res[0]) - want) < 1e-12, (trial, r, m, ld, mode, use_pose, float(res[0]), want, res)
    d = abs(got - want)
    worst, n = max(worst, d), n + 1
    assert d < 1e-12, (trial, r, m, ld, mode, use_pose, got, want)
print("fuzz ok: %d cases (%d took the multi-call path), worst |dF1| %.2e" % (n, fell_back, worst))
