#!/usr/bin/env python3
"""Launch latency of the embed kernel for few graphs (config-2 shape, node_num 64): usage lat.py [G ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from sg_pr_amd import engine, synth  # noqa: E402

sd = torch.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "model.pth"), map_location="cpu")
eng = engine.Engine(sd)
for g in [int(x) for x in sys.argv[1:]] or [64, 128, 256, 384, 512, 768, 1024]:
    c, l, _ = synth.make_graphs(g, 64, 20, 54, 0)
    order, cap = eng.size_order(c, l, 10)
    c, l = torch.from_numpy(c).cuda(), torch.from_numpy(l).cuda()
    for _ in range(200):
        eng.embed(c, l, 10, node_cap=cap, order=order)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(500):
        eng.embed(c, l, 10, node_cap=cap, order=order)
    e1.record()
    torch.cuda.synchronize()
    print("G %5d  %.2f us per embed call (kernel + second pass)" % (g, e0.elapsed_time(e1) * 2))
