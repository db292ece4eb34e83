#!/usr/bin/env python3
"""Experiment: the KITTI-00-sized embed as ONE ordered launch vs two concurrent launches on two streams (graphs that need
the 64-row layout / graphs that fit the 48-row layout at five workgroups per CU)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from sg_pr_amd import engine, synth  # noqa: E402

sd = torch.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "model.pth"), map_location="cpu")
eng = engine.Engine(sd)
c, l, _, _ = synth.kitti_like_sequence(4541, 100, 0)
eff = eng.processed_slots(c, l, 10)
order, cap = eng.size_order(c, l, 10)
big = order[: int((eff > 48).sum())]
small = order[int((eff > 48).sum()):]
print("graphs", len(order), "big", len(big), "small", len(small), "cap", cap)
dc, dl = torch.from_numpy(c).cuda(), torch.from_numpy(l).cuda()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def one():
    return eng.embed(dc, dl, 10, node_cap=cap, order=order)[0]


def two():
    ev = torch.cuda.Event()
    ev.record()
    sa.wait_event(ev)
    sb.wait_event(ev)
    with torch.cuda.stream(sa):
        pa = eng.embed(dc, dl, 10, node_cap=cap, order=big)[0]
    with torch.cuda.stream(sb):
        pb = eng.embed(dc, dl, 10, node_cap=48, order=small)[0]
    torch.cuda.current_stream().wait_stream(sa)
    torch.cuda.current_stream().wait_stream(sb)
    return pa, pb


def seq():
    pa = eng.embed(dc, dl, 10, node_cap=cap, order=big)[0]
    pb = eng.embed(dc, dl, 10, node_cap=48, order=small)[0]
    return pa, pb


for name, fn in (("one launch", one), ("two streams", two), ("two launches, one stream", seq), ("one launch", one)):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    print("%-28s %.1f us per embed" % (name, (time.perf_counter() - t0) / 200 * 1e6))
p1 = one()
pa, pb = two()
torch.cuda.synchronize()
print("bitwise:", torch.equal(p1[big.long()], pa[big.long()]), torch.equal(p1[small.long()], pb[small.long()]))
