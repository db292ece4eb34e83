import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from sg_pr_amd import engine, synth
sd = torch.load("tests/golden/model.pth", map_location="cpu")
eng = engine.Engine(sd)
c, l, _ = synth.config5_pairs(seed=0)
order, cap = eng.size_order(c, l, 20)
dc, dl = torch.from_numpy(c).cuda(), torch.from_numpy(l).cuda()
def run(mask):
    eng.set_skip_mask(mask)
    for _ in range(20): p = eng.embed(dc, dl, 20, node_cap=cap, order=order)[0]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): p = eng.embed(dc, dl, 20, node_cap=cap, order=order)[0]
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50 * 1e3
    eng.set_skip_mask(0)
    return dt, p
t0, p0 = run(256)
t1, p1 = run(256 + 65536)
print("bisect %.3f ms   networks %.3f ms   equal %s  maxdiff %g" % (t0, t1, torch.equal(p0, p1), (p0 - p1).abs().max().item()))
