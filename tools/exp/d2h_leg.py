# Experiment: where does the d2h leg of bench.py (H2D ragged graphs + embed + all-pairs + D2H of the matrix) lose time against
# the bare copy (1.46 ms for 82.5 MB)?  Variants of the copy-out.
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from sg_pr_amd import engine, synth
sd = torch.load("tests/golden/model.pth", map_location="cpu")
eng = engine.Engine(sd)
dev = eng.device
c, l, _, _ = synth.kitti_like_sequence(4541, 100, 0)
rag = eng.to_ragged(c, l)
pinned = tuple(torch.from_numpy(x).pin_memory() for x in rag)
order, cap = eng.ragged_order(rag[2], 100, 10)
m = 4541
ho = torch.empty(m, m, dtype=torch.float32).pin_memory()
do = torch.empty(m, m, dtype=torch.float32, device=dev)
copy_stream = torch.cuda.Stream(device=dev)
def leg(mode, pieces=4):
    dc, dl, do_ = (x.to(dev, non_blocking=True) for x in pinned)
    pooled = eng.embed_ragged(dc, dl, do_, 100, 10, node_cap=cap, order=order)[0]
    if mode == "nocopy":
        eng.score_all_pairs(pooled, pooled, out=do)
    elif mode == "after":
        eng.score_all_pairs(pooled, pooled, out=do)
        ho.copy_(do, non_blocking=True)
    elif mode == "pieces_same_stream":
        for q in range(pieces):
            r0, r1 = m * q // pieces, m * (q + 1) // pieces
            eng.score_all_pairs(pooled[r0:r1].contiguous(), pooled, out=do[r0:r1])
            ho[r0:r1].copy_(do[r0:r1], non_blocking=True)
    else:
        for q in range(pieces):
            r0, r1 = m * q // pieces, m * (q + 1) // pieces
            eng.score_all_pairs(pooled[r0:r1].contiguous(), pooled, out=do[r0:r1])
            ready = torch.cuda.Event(); ready.record()
            copy_stream.wait_event(ready)
            with torch.cuda.stream(copy_stream):
                ho[r0:r1].copy_(do[r0:r1], non_blocking=True)
    torch.cuda.synchronize()
def prewarm(sec=1.5):
    t0 = time.perf_counter()
    p = eng.embed(torch.from_numpy(c).to(dev), torch.from_numpy(l).to(dev), 10)[0]
    while time.perf_counter() - t0 < sec:
        for _ in range(50): eng.score_all_pairs(p, p, out=do)
        torch.cuda.synchronize()
prewarm()
def t(mode, **kw):
    for _ in range(10): leg(mode, **kw)
    t0 = time.perf_counter()
    for _ in range(40): leg(mode, **kw)
    return (time.perf_counter() - t0) / 40 * 1e3
for mode, kw in (("nocopy", {}), ("after", {}), ("pieces_same_stream", {}), ("pieces", {"pieces": 4}), ("pieces", {"pieces": 2}), ("pieces", {"pieces": 8}), ("pieces", {"pieces": 16})):
    print("%-22s %-14s %.3f ms" % (mode, kw, t(mode, **kw)))
