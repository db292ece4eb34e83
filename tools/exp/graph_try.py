# Experiment (round 2): replaying one KITTI-00 step as a captured hipGraph (torch.cuda.CUDAGraph around the ctypes launches)
# against eager launches: 0.317 vs 0.312 ms per step - the step has no launch gaps to recover, so bench.py stays eager.
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
from sg_pr_amd import sg_net, synth, allpairs
from sg_pr_amd.parser_sg import sgpr_args
args = sgpr_args(); args.model = "tests/golden/model.pth"
tr = sg_net.SGTrainer(args, False); model = tr.model; eng = model.engine()
c, l, _, _ = synth.kitti_like_sequence(4541, 100, 0)
order, cap = eng.size_order(c, l, 10)
dc, dl = torch.from_numpy(c).cuda(), torch.from_numpy(l).cuda()
sc = allpairs.AllPairsScorer(model=model)
sc.embed_fn = lambda cc, ll: eng.embed(cc, ll, 10, node_cap=cap, order=order)[0]
def step(): return sc.run(dc, dl)
for _ in range(300): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300): m = step()
torch.cuda.synchronize(); print("eager ms/step", (time.perf_counter() - t0) / 300 * 1e3)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    out = step()
torch.cuda.synchronize()
for _ in range(50): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300): g.replay()
torch.cuda.synchronize(); print("graph ms/step", (time.perf_counter() - t0) / 300 * 1e3)
print("equal", torch.equal(out, m))
