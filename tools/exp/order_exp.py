# Experiment: does the launch ORDER of sgpr_embed_ordered matter beyond "largest first"?  (KITTI-00-like bench data)
#   a  largest first (production)        b  random          c  big / small interleaved (LPT within pairs)
#   d  largest first, but the first 1024 launch slots shuffled (phases of co-resident workgroups desynchronised)
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from sg_pr_amd import engine, synth
sd = torch.load("tests/golden/model.pth", map_location="cpu")
eng = engine.Engine(sd)
c, l, _, _ = synth.kitti_like_sequence(4541, 100, 0)
dc, dl = torch.from_numpy(c).cuda(), torch.from_numpy(l).cuda()
order, cap = eng.size_order(dc, dl, 10)
o = order.cpu().numpy()
rng = np.random.default_rng(0)
G = o.size
inter = np.empty(G, dtype=np.int32); inter[0::2] = o[: (G + 1) // 2]; inter[1::2] = o[(G + 1) // 2:][::-1]
d = o.copy(); d[:1024] = rng.permutation(d[:1024])
e = o.copy()
for s in range(0, G, 1024): e[s:s + 1024] = rng.permutation(e[s:s + 1024])
orders = {"largest first": o, "random": rng.permutation(o), "big/small interleaved": inter, "first 1024 shuffled": d, "shuffled within rounds of 1024": e,
          "smallest first": o[::-1].copy()}
def run(od):
    od = torch.from_numpy(np.ascontiguousarray(od)).cuda()
    for _ in range(20): p = eng.embed(dc, dl, 10, node_cap=cap, order=od)[0]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(100): p = eng.embed(dc, dl, 10, node_cap=cap, order=od)[0]
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 10.0, p
ref = None
for rep in range(2):
    for name, od in orders.items():
        t, p = run(od)
        if ref is None: ref = p
        print("%-32s %.1f us per launch   equal %s" % (name, t, torch.equal(p, ref)))
