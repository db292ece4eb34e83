import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from sg_pr_amd import sg_net, synth
from sg_pr_amd.parser_sg import sgpr_args
args = sgpr_args()
args.filters_1, args.filters_2, args.filters_3, args.tensor_neurons, args.bottle_neck_neurons = 128, 128, 64, 32, 32
args.node_num, args.K = 100, 10
torch.manual_seed(1)
model = sg_net.SG(args, 12).eval()
eng = model.engine()
G = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
c, l, _, _ = synth.kitti_like_sequence(G, 100, 0)
lab = np.where(l >= 0, l % 12, l).astype(np.int32)
cg, lg = torch.from_numpy(c).cuda(), torch.from_numpy(lab).cuda()
for _ in range(5):
    p = eng.embed(cg, lg, 10)[0]
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    p = eng.embed(cg, lg, 10)[0]
e1.record(); torch.cuda.synchronize()
print("G %d: %.1f us per call = %.3f us per graph" % (G, e0.elapsed_time(e1) * 100, e0.elapsed_time(e1) * 100 / G))
