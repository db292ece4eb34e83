#!/usr/bin/env python3
"""Fuzz: HIP engine vs the CPU oracle on random shapes (node_num, K, node counts): scores within 1e-4."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sg_pr_amd import engine, synth
from oracle import sgpr_oracle as oracle
sd = torch.load("tests/golden/model.pth", map_location="cpu")
osd = oracle.load_checkpoint("tests/golden/model.pth")
eng = engine.Engine(sd)
rng = np.random.default_rng(2024)
worst, worst_att, flips, worst_list = 0.0, 0.0, 0, 0.0
for trial in range(40):
    n = int(rng.integers(17, 257))
    k = int(rng.integers(1, min(32, n // 2) + 1))
    hi = n - k
    lo = int(rng.integers(1, hi + 1))
    g = 8
    c, l, _ = synth.make_graphs(g, n, lo, hi, int(rng.integers(1 << 30)), kitti_like=bool(rng.integers(2)))
    dense = torch.from_numpy(synth.dense_features(c, l))
    rp, ra, _ = oracle.embed(osd, dense, k)
    for mode in ("packed", "dense", "lean", "ragged"):
        if mode == "packed":
            p, a, _ = eng.embed(c, l, k, want_att=True)
        elif mode == "ragged":
            rc_, rl_, off_ = eng.to_ragged(c, l)
            order_r, cap_r = eng.ragged_order(off_, n, k)
            p, a, _ = eng.embed_ragged(rc_, rl_, off_, n, k, want_att=True, node_cap=cap_r, order=order_r)
            p0 = eng.embed(c, l, k)[0]
            assert torch.equal(p, p0), "ragged store differs from the padded arrays (n=%d k=%d)" % (n, k)
        elif mode == "dense":
            p, a, _ = eng.embed_dense(dense, k, want_att=True)
        else:
            order, cap = eng.size_order(np.tile(c, (40, 1, 1)), np.tile(l, (40, 1)), k)
            p, a, _ = eng.embed(np.tile(c, (40, 1, 1)), np.tile(l, (40, 1)), k, want_att=True, node_cap=cap, order=order)
            p, a = p[:g], a[:g]
        s = eng.score_all_pairs(p, p).cpu()
        rs = oracle.score_all_pairs(osd, rp, rp)
        d = (s - rs).abs().max().item()
        ii, jj = torch.meshgrid(torch.arange(g, dtype=torch.int32), torch.arange(g, dtype=torch.int32), indexing="ij")
        sl = eng.score_pairs(p, p, ii.reshape(-1), jj.reshape(-1)).view(g, g).cpu()
        worst_list = max(worst_list, (sl - rs).abs().max().item())
        if d > 2e-5:
            print("score dev %.2e all-pairs, %.2e pair-list kernel (n=%d k=%d mode=%s) |z| where: score=%.4f" % (d, (sl - rs).abs().max().item(), n, k, mode, float(rs.flatten()[(s - rs).abs().argmax()])))
        da = (a.cpu() - ra.squeeze(-1)).abs().max().item()
        worst, worst_att = max(worst, d), max(worst_att, da)
        if da > 1e-4:
            flips += 1
            print("att deviation %.2e (n=%d k=%d lo=%d mode=%s) score dev %.2e" % (da, n, k, lo, mode, d))
eng.check_status()
print("40 shapes x 4 entry points: max |dscore| all-pairs %.3e / pair-list %.3e, max |datt| %.3e, shapes with |datt| > 1e-4: %d" % (worst, worst_list, worst_att, flips))
