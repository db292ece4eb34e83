#!/usr/bin/env python3
"""All-pairs tail with a given leading dimension of the output (target for rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from sg_pr_amd import engine, synth  # noqa: E402
ld = int(sys.argv[1])
sd = torch.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "model.pth"), map_location="cpu")
eng = engine.Engine(sd)
c, l, _, _ = synth.kitti_like_sequence(4541, 100, 0)
order, cap = eng.size_order(c, l, 10)
p = eng.embed(torch.from_numpy(c).cuda(), torch.from_numpy(l).cuda(), 10, node_cap=cap, order=order)[0]
buf = torch.empty(4541, ld, device="cuda")
for _ in range(3):
    m = eng.score_all_pairs(p, p, out=buf[:, :4541])
torch.cuda.synchronize()
print("ok", float(m[0, 0]))
