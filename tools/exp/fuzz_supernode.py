#!/usr/bin/env python3
"""Fuzz: production embed (super-node semantic branch) vs the generic branch (ablation bit 12) - bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sg_pr_amd import engine, synth
eng = engine.Engine(torch.load("tests/golden/model.pth", map_location="cpu"))
rng = np.random.default_rng(123)
bad = 0
cases = 0
for trial in range(60):
    n = int(rng.choice([33, 40, 64, 100, 128, 160]))
    k = int(rng.choice([1, 2, 3, 5, 10, 16, 20, 32]))
    if k >= n // 2:
        k = 10
    lo = int(rng.integers(1, max(2, n - k - 2)))
    hi = int(rng.integers(lo, n - k + 1))
    g = int(rng.choice([7, 300]))
    c, l, _ = synth.make_graphs(g, n, lo, hi, int(rng.integers(1 << 30)), kitti_like=bool(rng.integers(2)))
    if trial % 5 == 0:                       # few labels only: large label groups (>= k mates)
        l = np.where(l >= 0, l % 2, l).astype(np.int32)
        for q in range(g):
            l[q, :(l[q] >= 0).sum()].sort()
    order, cap = eng.size_order(c, l, k)
    fast = eng.embed(c, l, k, want_att=True, node_cap=cap, order=order)
    eng.set_skip_mask(256 + 4096)
    gen = eng.embed(c, l, k, want_att=True, node_cap=cap, order=order)
    eng.set_skip_mask(0)
    eng.check_status()
    ok = torch.equal(fast[0], gen[0]) and torch.equal(fast[1], gen[1])
    cases += 1
    if not ok:
        bad += 1
        d = (fast[0] - gen[0]).abs().max().item()
        print("MISMATCH n=%d k=%d lo=%d hi=%d g=%d max|d pooled|=%g" % (n, k, lo, hi, g, d))
print("cases", cases, "mismatches", bad)
