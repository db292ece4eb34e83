#!/usr/bin/env python3
"""Per-phase cycle split of the embed kernel on the three benchmark shapes (GPU box only)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from sg_pr_amd import engine, synth  # noqa: E402

sd = torch.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "model.pth"), map_location="cpu")
eng = engine.Engine(sd)
for name, (n, k, g) in {"kitti00": (100, 10, 4541), "pairs128": (64, 10, 256), "stress": (256, 20, 512)}.items():
    if name == "kitti00":
        c, l, _, _ = synth.kitti_like_sequence(g, n, 0)
    else:
        c, l, _ = synth.make_graphs(g, n, n // 3, n - k, 0)
    order, cap = eng.size_order(c, l, k)
    c, l = torch.from_numpy(c).cuda(), torch.from_numpy(l).cuda()
    eng.embed(c, l, k, node_cap=cap, order=order)
    frac, cyc = eng.phase_profile(c, l, k, node_cap=cap, order=order)
    print(name, "N=%d k=%d G=%d" % (n, k, g), " ".join("%s=%.1f%%" % (p, 100 * f) for p, f in frac.items() if p != "-"),
          "| cycles/graph=%.0f  launch %.3f ms with timers" % (cyc.sum() / 3 / g, eng.last_profile_ms))
    print("   gemm split (first weights, a-tiles, b-tiles, barrier):", (eng.last_select_split[:4] / 3 / g).round())
    print("   cycles per graph:", " ".join("%s=%.0f" % (p, v / 3 / g) for p, v in zip(eng.PHASES, cyc) if p != "-"))
