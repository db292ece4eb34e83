#!/usr/bin/env python3
"""Plain sgpr_embed (no node_cap promise: lean plan + hand-over of oversize graphs) against the capped / ordered launches:
bitwise equality and time per launch (HIP events around `reps` back-to-back calls)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from sg_pr_amd import engine, synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
sd = torch.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "model.pth"), map_location="cpu")
eng = engine.Engine(sd)


def timed(fn):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        out = fn()
    b.record()
    torch.cuda.synchronize()
    return out, a.elapsed_time(b) / reps * 1e3


cases = []
c, l, _, _ = synth.kitti_like_sequence(4541, 100, 0)
cases.append(("kitti00 (<= 61 slots)", c, l, 10))
c, l, _ = synth.make_graphs(4541, 100, 25, 70, seed=3, kitti_like=True)
cases.append(("n100, 25..70 nodes (some > 64)", c, l, 10))
c, l, _ = synth.make_graphs(4541, 100, 40, 85, seed=4)
cases.append(("n100, 40..85 nodes (most > 64)", c, l, 10))
c, l, _ = synth.config5_pairs(seed=0)
cases.append(("stress n256 k20", c, l, 20))
c, l, _ = synth.make_graphs(2048, 256, 20, 120, seed=5)
cases.append(("n256 k10, 20..120 nodes", c, l, 10))
for name, c, l, k in cases:
    order, cap = eng.size_order(c, l, k)
    eff = eng.processed_slots(c, l, k)
    dc, dl = torch.from_numpy(c).cuda(), torch.from_numpy(l).cuda()
    plain, t_plain = timed(lambda: eng.embed(dc, dl, k, auto_order=False)[0])
    eng.check_status()
    capped, t_cap = timed(lambda: eng.embed(dc, dl, k, node_cap=cap, auto_order=False)[0])
    ordered, t_ord = timed(lambda: eng.embed(dc, dl, k, node_cap=cap, order=order)[0])
    default, t_def = timed(lambda: eng.embed(dc, dl, k)[0])           # the binding's default: plain call + cached device order
    eng.check_status()
    print("%-34s cap %3d  >64: %4d  plain %8.1f us  capped %8.1f  ordered %8.1f  engine default %8.1f  plain/ordered %.3f  default/ordered %.3f  bitwise %s" % (
        name, cap, int((eff > 64).sum()), t_plain, t_cap, t_ord, t_def, t_plain / t_ord, t_def / t_ord,
        bool(torch.equal(plain, capped) and torch.equal(plain, ordered) and torch.equal(plain, default))))
