#!/usr/bin/env python3
"""Whole-sequence parity census: embed every graph of the KITTI-00-sized synthetic sequence with the engine and with the
CPU oracle, and count the graphs whose embedding differs by more than rounding (a neighbour chosen differently between
two near-tied candidates), the layer it happened in, and what that does to the score matrix.
usage: [SEQ_PARITY_MASK=8192] seq_parity.py [num_graphs=4541]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sg_pr_amd import engine, synth
from oracle import sgpr_oracle as oracle
G = int(sys.argv[1]) if len(sys.argv) > 1 else 4541
sd = torch.load("tests/golden/model.pth", map_location="cpu")
osd = oracle.load_checkpoint("tests/golden/model.pth")
eng = engine.Engine(sd)
if os.environ.get("SEQ_PARITY_MASK"):
    eng.set_skip_mask(int(os.environ["SEQ_PARITY_MASK"]))        # 8192: the wide-range (three bf16 planes, 24 bits) instance
c, l, _, poses = synth.kitti_like_sequence(G, 100, seed=0)
t0 = time.time()
rp = []
torch.set_num_threads(32)
for s in range(0, G, 256):
    dense = torch.from_numpy(synth.dense_features(c[s:s + 256], l[s:s + 256]))
    rp.append(oracle.embed(osd, dense, 10)[0])
rp = torch.cat(rp)
print("oracle: %d graphs in %.1f s" % (G, time.time() - t0))
p, att, _ = eng.embed(c, l, 10, want_att=True)
p = p.cpu()
dev = (p - rp).abs().amax(1)
print("max|d pooled| percentiles 50/99/99.9/max: %.2e %.2e %.2e %.2e" % tuple(np.quantile(dev.numpy(), [0.5, 0.99, 0.999, 1.0])))
flipped = np.flatnonzero(dev.numpy() > 2e-4)
print("graphs with |d pooled| > 2e-4 (a differently chosen neighbour):", len(flipped), "of", G, flipped[:20].tolist())
s = eng.score_all_pairs(eng_p := p.cuda(), eng_p).cpu()
rs = oracle.score_all_pairs(osd, rp, rp)
d = (s - rs).abs()
print("score matrix: max |d| %.3e, pairs with |d| > 1e-4: %d of %d (%.2e), > 1e-3: %d" % (d.max().item(), int((d > 1e-4).sum()), d.numel(), float((d > 1e-4).float().mean()), int((d > 1e-3).sum())))
clean = np.setdiff1d(np.arange(G), flipped)
print("score matrix without those graphs: max |d| %.3e" % d[clean][:, clean].max().item())
