#!/usr/bin/env python3
"""Ablation timing of the embed kernel: wall time per launch with phases skipped (GPU box only)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from sg_pr_amd import engine, synth  # noqa: E402

sd = torch.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "model.pth"), map_location="cpu")
eng = engine.Engine(sd)
shapes = {"kitti00": (100, 10, 4541), "pairs128": (64, 10, 256)}
for name, (n, k, g) in shapes.items():
    if name == "kitti00":
        c, l, _, _ = synth.kitti_like_sequence(g, n, 0)
    else:
        c, l, _ = synth.make_graphs(g, n, 20, n - k, 0)
    order, cap = eng.size_order(c, l, k)
    c, l = torch.from_numpy(c).cuda(), torch.from_numpy(l).cuda()
    for mask, label in [(0, "full"), (256, "full, profile instance"), (256 + 4096, "full, generic first semantic layer"), (1, "no select"), (2, "no gemm"), (512, "gemm without weight loads"), (1024, "gemm without MFMAs"), (2048, "gemm without its inner barrier"), (1024+512, "gemm: no weights, no MFMAs"), (3, "no select, no gemm"), (4, "no gram"),
                        (8, "no gather"), (15, "skeleton (stage, barriers, conv_end, attention)"),
                        (32, "input fetch + duplicate detection only"), (16, "dispatch only")]:
        eng.set_skip_mask(mask)
        for _ in range(3):
            eng.embed(c, l, k, node_cap=cap, order=order)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            eng.embed(c, l, k, node_cap=cap, order=order)
        torch.cuda.synchronize()
        print("%-9s %-48s %.4f ms" % (name, label, (time.perf_counter() - t0) * 100))
    eng.set_skip_mask(0)
