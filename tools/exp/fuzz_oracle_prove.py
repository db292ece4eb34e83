#!/usr/bin/env python3
"""The shapes of tools/exp/fuzz_oracle.py whose scores or attention leave rounding level: is every such graph a proven
near-tie (tests/tie_proof.py), or a bug?   python tools/exp/fuzz_oracle_prove.py"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np, torch
from sg_pr_amd import engine, synth
from oracle import sgpr_oracle as oracle
import tie_proof
sd = torch.load(os.path.join(root, "tests/golden/model.pth"), map_location="cpu")
osd = oracle.load_checkpoint(os.path.join(root, "tests/golden/model.pth"))
eng = engine.Engine(sd)
rng = np.random.default_rng(2024)
bad = 0
for trial in range(40):
    n = int(rng.integers(17, 257))
    k = int(rng.integers(1, min(32, n // 2) + 1))
    hi = n - k
    lo = int(rng.integers(1, hi + 1))
    g = 8
    c, l, _ = synth.make_graphs(g, n, lo, hi, int(rng.integers(1 << 30)), kitti_like=bool(rng.integers(2)))
    dense = torch.from_numpy(synth.dense_features(c, l))
    rp, ra, _ = oracle.embed(osd, dense, k)
    p, a, _ = eng.embed(c, l, k, want_att=True)
    dp = (p.cpu() - rp).abs().amax(1)
    for gi in np.nonzero((dp > 2e-4).numpy())[0]:
        rep = tie_proof.prove_graph(eng, oracle, osd, c[gi], l[gi], k, pooled_g=p[gi].cpu())
        print("trial %d n=%d k=%d graph %d: |d pooled| %.2e -> %s %s" % (trial, n, k, gi, float(dp[gi]),
              "proven near-tie" if rep["proven"] else "NOT PROVEN", {kk: v for kk, v in rep.items() if kk in ("reason", "flips", "layer", "worst_gap")}))
        bad += 0 if rep["proven"] else 1
print("unproven graphs:", bad)
