#!/usr/bin/env python3
"""Fuzz of the device F1-max / ROC-area path against the sorted host computation: random shapes, label densities, tie
structures (rounded scores, saturated scores), row shards and padded leading dimensions."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from sg_pr_amd import engine, metrics  # noqa: E402

sd = torch.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "model.pth"), map_location="cpu")
eng = engine.Engine(sd)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst_f, worst_a, n = 0.0, 0.0, 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    r, m = int(rng.integers(1, 900)), int(rng.integers(1, 1500))
    ld = m + int(rng.integers(0, 9))
    p_pos = float(rng.choice([0.001, 0.01, 0.1, 0.5, 0.9]))
    lab = np.where(rng.random((r, m)) < 0.1, -1, (rng.random((r, m)) < p_pos).astype(np.int8)).astype(np.int8)
    z = rng.normal(0, 3, (r, m)) + 3.0 * (lab == 1)
    sc = (1 / (1 + np.exp(-z))).astype(np.float32)
    mode = trial % 4
    if mode == 1:
        sc = np.round(sc, 2).astype(np.float32)
    elif mode == 2:
        sc = np.where(rng.random((r, m)) < 0.3, np.float32(1.0), sc).astype(np.float32)
    elif mode == 3:
        sc = np.round(sc, 4).astype(np.float32)
    buf = torch.zeros(r, ld, device="cuda")
    buf[:, :m] = torch.from_numpy(sc).cuda()
    keep = lab.ravel() >= 0
    if not (lab == 1).any():
        continue
    f, a, passes = metrics.pr_roc_device(eng, buf[:, :m], gt=torch.from_numpy(lab))
    wf = metrics.f1_max(lab.ravel()[keep], sc.ravel()[keep])
    wa = metrics.roc_auc(lab.ravel()[keep], sc.ravel()[keep])
    df = abs(f - wf)
    da = 0.0 if (np.isnan(a) and np.isnan(wa)) else abs(a - wa)
    worst_f, worst_a, n = max(worst_f, df), max(worst_a, da), n + 1
    assert df < 1e-12 and da < 1e-12, (trial, r, m, ld, p_pos, mode, f, wf, a, wa, passes)
print("fuzz ok:", n, "cases, worst |dF1| %.2e |dAUC| %.2e" % (worst_f, worst_a))
