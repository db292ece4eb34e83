#!/usr/bin/env python3
"""The KITTI-00 step (ordered embed + dense all-pairs tail) issued eagerly against a captured hipGraph replayed per step:
what the launch gaps between its four kernels cost (HIP events around `reps` back-to-back steps)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from sg_pr_amd import engine, synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
sd = torch.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "model.pth"), map_location="cpu")
eng = engine.Engine(sd)
c, l, _, _ = synth.kitti_like_sequence(4541, 100, 0)
dc, dl = torch.from_numpy(c).cuda(), torch.from_numpy(l).cuda()
order, cap = eng.size_order(dc, dl, 10)
out = torch.empty(4541, 4541, device="cuda")


def step():
    p = eng.embed(dc, dl, 10, node_cap=cap, order=order)[0]
    return eng.score_all_pairs(p, p, out=out)


def timed(fn):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for _ in range(300):
    step()
torch.cuda.synchronize()
ref = step().clone()
t_eager = timed(step)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    res = step()
g.replay()
torch.cuda.synchronize()
print("graph replay equals the eager step:", bool(torch.equal(res, ref)))
t_graph = timed(g.replay)
t_eager2 = timed(step)
print("eager %.4f ms  graph replay %.4f ms  eager again %.4f ms  (%.1f %%)" % (t_eager, t_graph, t_eager2, 100 * (t_graph / min(t_eager, t_eager2) - 1)))
