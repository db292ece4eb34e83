# Experiment: graphs of <= 48 processed slots on the 48-row lean layout (five workgroups per CU) against the 64-row one
# (round 2, 4541 graphs of 25..47 nodes: 0.1665 ms on the 48-row layout vs 0.1833 ms on the 64-row one.  Splitting a mixed
#  data set into a > 48 and a <= 48 launch was slower than one launch: 0.212 vs 0.199 ms - two launch tails.)
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from sg_pr_amd import engine, synth
sd = torch.load("tests/golden/model.pth", map_location="cpu")
eng = engine.Engine(sd)
c, l, _ = synth.make_graphs(4541, 100, 25, 47, 0, kitti_like=True)
order, cap = eng.size_order(c, l, 10)
dc, dl = torch.from_numpy(c).cuda(), torch.from_numpy(l).cuda()
def run(cap_):
    for _ in range(300): p = eng.embed(dc, dl, 10, node_cap=cap_, order=order)[0]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): p = eng.embed(dc, dl, 10, node_cap=cap_, order=order)[0]
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 300 * 1e3, p
print("cap", cap)
t48, p48 = run(cap); t64, p64 = run(64); t48b, _ = run(cap)
print("48-row layout %.4f / %.4f ms   64-row layout %.4f ms   equal %s" % (t48, t48b, t64, torch.equal(p48, p64)))
