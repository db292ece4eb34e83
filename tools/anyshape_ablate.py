#!/usr/bin/env python3
"""Where the any-shape embed kernel's time goes: its phases skipped one at a time (debug skip mask bits 24..27: kNN dot
products, selection bisection, the a / b products, the gather) - the outputs are wrong, the times tell.  KITTI-00 shape."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from sg_pr_amd import engine, synth  # noqa: E402

sd = torch.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "model.pth"), map_location="cpu")
sd = {k[7:] if k.startswith("module.") else k: v for k, v in sd.items()}
w = sd["dgcnn_f_conv1.0.weight"]
sd["dgcnn_f_conv1.0.weight"] = torch.cat((w.reshape(w.shape[0], 2, 12), torch.zeros(w.shape[0], 2, 1)), dim=2).reshape(w.shape[0], 26, 1, 1)
eng = engine.Engine(sd, engine.SgprDims(13, 64, 64, 32, 16, 16))
c, l, _, _ = synth.kitti_like_sequence(4541, 100, 0)
cd, ld = torch.from_numpy(c).cuda(), torch.from_numpy(l).cuda()
for name, mask in (("all phases", 0), ("no kNN dot products", 1), ("no bisection", 2), ("no a / b products", 4), ("no gather", 8),
                   ("none of the four", 15)):
    eng.set_skip_mask(mask << 24)
    eng.embed(cd, ld, 10)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3):
        eng.embed(cd, ld, 10)
    b.record()
    torch.cuda.synchronize()
    print("%-24s %8.1f us" % (name, a.elapsed_time(b) / 3 * 1e3))
