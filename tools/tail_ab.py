#!/usr/bin/env python3
"""The all-pairs tail of a KITTI-00-sized matrix through sgpr_score_all_pairs and through the multi-rectangle entry with
one job (same rectangle), event-timed; and the five KITTI matrices through the multi entry.  Same-box A/B helper."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sg_pr_amd import engine, synth
sd = torch.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "model.pth"), map_location="cpu")
eng = engine.Engine(sd)
def ms(fn, reps=40):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
pl, outs = [], []
for si, m in enumerate((4541, 4661, 2761, 1101, 4071)):
    c, l, _, _ = synth.kitti_like_sequence(m, 100, si)
    order, cap = eng.size_order(c, l, 10)
    p = eng.embed(c, l, 10, node_cap=cap, order=order)[0]
    pl.append(p); outs.append(torch.empty(m, m, dtype=torch.float32, device=p.device))
for _ in range(200): eng.score_all_pairs(pl[0], pl[0], out=outs[0])     # clock ramp
a = ms(lambda: eng.score_all_pairs(pl[0], pl[0], out=outs[0]))
ref = outs[0].clone()
b = ms(lambda: eng.score_all_pairs_multi([(pl[0], pl[0], outs[0])]))
same = bool(torch.equal(ref, outs[0]))
c5 = ms(lambda: eng.score_all_pairs_multi([(p, p, o) for p, o in zip(pl, outs)]), 20)
s5 = ms(lambda: [eng.score_all_pairs(p, p, out=o) for p, o in zip(pl, outs)], 20)
print("kitti00 tail call: single entry %.1f us, multi entry with one job %.1f us (bit-identical: %s); five matrices: multi %.1f us, five single calls %.1f us"
      % (a * 1e3, b * 1e3, same, c5 * 1e3, s5 * 1e3))
