#!/usr/bin/env python3
"""Fuzz of sgpr_pair_plan + sgpr_score_pair_list against the dense rectangle (bit for bit): random rectangles, list
lengths from 1 to 60 k, row distributions (uniform, one hot row, Zipf-like, every row once, sorted / shuffled lists,
repeated pairs), pooled vectors inside and outside the f16 range.   python fuzz_pair_list.py seed trials"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from sg_pr_amd import engine, synth  # noqa: E402

sd = torch.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "model.pth"), map_location="cpu")
eng = engine.Engine(sd)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 60
c, l, _ = synth.make_graphs(1500, 100, 20, 60, seed=3, kitti_like=True)
pool = eng.embed(c, l, 10)[0]
n = 0
for trial in range(trials):
    r, m = int(rng.integers(1, 1500)), int(rng.integers(1, 1500))
    rows = pool[torch.from_numpy(rng.permutation(1500)[:r]).cuda()].contiguous()
    cols = pool[torch.from_numpy(rng.permutation(1500)[:m]).cuda()].contiguous()
    if trial % 7 == 6:
        rows, cols = rows * 300.0, cols * 300.0               # outside the f16 range: the exact fp32 path
    p = int(rng.choice([1, 2, 15, 16, 17, 100, 2047, 2048, 5000, 60000]))
    mode = trial % 5
    if mode == 0:
        i1 = rng.integers(0, r, p)
    elif mode == 1:
        i1 = np.full(p, int(rng.integers(0, r)))
    elif mode == 2:
        i1 = np.minimum((rng.pareto(1.2, p) * 3).astype(np.int64), r - 1)
    elif mode == 3:
        i1 = np.arange(p) % r
    else:
        i1 = np.sort(rng.integers(0, r, p))
    i2 = rng.integers(0, m, p)
    if trial % 4 == 0 and p > 4:
        i2[: p // 2] = i2[0]                                   # repeated columns / pairs
    plan = eng.pair_plan(i1, i2, r, m)
    got = eng.score_pair_list(rows, cols, plan)
    dense = eng.score_all_pairs(rows, cols)
    want = dense[torch.from_numpy(i1).cuda(), torch.from_numpy(i2).cuda()]
    assert torch.equal(got, want), (trial, r, m, p, mode, (got - want).abs().max().item())
    n += 1
print("fuzz ok: %d pair lists bit-identical to the dense rectangle" % n)
