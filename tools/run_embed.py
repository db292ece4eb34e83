#!/usr/bin/env python3
"""Run the embed kernel a few times on one benchmark shape (target for rocprofv3 --pmc)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from sg_pr_amd import engine, synth  # noqa: E402

shape = sys.argv[1] if len(sys.argv) > 1 else "kitti00"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
mask = int(sys.argv[3]) if len(sys.argv) > 3 else 0     # ablation mask (sgpr_debug_set_skip_mask)
n, k, g = {"kitti00": (100, 10, 4541), "pairs128": (64, 10, 256), "stress": (256, 20, 2048)}[shape]
sd = torch.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "model.pth"), map_location="cpu")
eng = engine.Engine(sd)
if shape == "kitti00":
    c, l, _, _ = synth.kitti_like_sequence(g, n, 0)
elif shape == "stress":
    c, l, _ = synth.config5_pairs(seed=0)          # the bench's --workload stress input
else:
    c, l, _ = synth.config2_pairs(seed=0)          # the bench's --workload pairs128 input
order, cap = eng.size_order(c, l, k)
eng.set_skip_mask(mask)
c, l = torch.from_numpy(c).cuda(), torch.from_numpy(l).cuda()
for _ in range(reps):
    p = eng.embed(c, l, k, node_cap=cap, order=order)[0]
    m = eng.score_all_pairs(p, p) if (shape == "kitti00" and mask == 0) else eng.score_pairs(p[0::2].contiguous(), p[1::2].contiguous())
torch.cuda.synchronize()
print("ok", float(p.sum()))
