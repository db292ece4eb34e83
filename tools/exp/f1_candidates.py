#!/usr/bin/env python3
"""Where do sgpr_f1_max's candidate bins lie, and how many (4 rows x 256 columns) blocks of the matrix hold a negative in
one of them?  (what a block-level filter of pass B could skip)   usage: f1_candidates.py [kitti|world]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sg_pr_amd import allpairs, engine, synth
kind = sys.argv[1] if len(sys.argv) > 1 else "kitti"
sd = torch.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "model.pth"), map_location="cpu")
eng = engine.Engine(sd)
gen = synth.world_sequence if kind == "world" else synth.kitti_like_sequence
c, l, _, poses = gen(4541, 100, seed=0)
order, cap = eng.size_order(c, l, 10)
p = eng.embed(torch.from_numpy(c).cuda(), torch.from_numpy(l).cuda(), 10, node_cap=cap, order=order)[0]
mat = eng.score_all_pairs(p, p)
xz = allpairs.pose_xz(poses).cuda()
lib, h = eng.lib, eng._h
r, m = mat.shape
ws_bytes = lib.sgpr_f1_max_workspace_bytes(h, r, m)
ws = torch.zeros(ws_bytes, dtype=torch.uint8, device="cuda")
res = torch.empty(8, dtype=torch.float64, device="cuda")
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
rc = lib.sgpr_f1_max(h, ctypes.c_void_p(mat.data_ptr()), r, m, mat.stride(0), 0, ctypes.c_void_p(xz.data_ptr()), 3.0, 20.0, None, m,
                     ctypes.c_void_p(res.data_ptr()), ctypes.c_void_p(ws.data_ptr()), ws_bytes, stream)
assert rc == 0
torch.cuda.synchronize()
print(kind, "result", res.cpu().tolist())
SH, HALF = 17, 0x3F000000 >> 17
ONE = 2 * HALF
NB = ONE + 1 + ((0x7F800000 - 0x3F800000) >> SH) + 1
NBP = (NB + 3) & ~3
a256 = lambda x: (x + 255) & ~255
PER = (NBP + 1023) // 1024
off = 256
off += a256(1024 * PER * 4)
off += a256((NBP + 4) * 8)
off += a256(1024 * PER * 8)
off += a256((4096 + 4) * 8)          # neg2
off += a256((NBP + 4) * 8)
off += a256((NBP + 4) * 8)
off_mark = off; off += a256((NBP // 32 + 1) * 4)
off_thr = off
T = int(res[6].item())
mark = ws[off_mark:off_mark + (NBP // 32 + 1) * 4].view(torch.int32).cpu().numpy().view(np.uint32)
thr = ws[off_thr:off_thr + 4 * T].view(torch.float32).cpu().numpy()
bins = np.nonzero(np.unpackbits(mark.view(np.uint8), bitorder="little"))[0]
print("thresholds", T, "min %.3e max %.3e" % (thr.min(), thr.max()), "| marked bins", len(bins), "from", bins.min(), "to", bins.max(), "of", NB)
bits = mat.view(torch.int32)
low = bits < 0x3F000000
key = torch.where(low, bits >> SH, ONE - ((1.0 - mat).view(torch.int32) >> SH))
marked = torch.zeros(NBP + 32, dtype=torch.bool, device="cuda")
marked[torch.from_numpy(bins.astype(np.int64)).cuda()] = True
hit = marked[key.long()]
d = torch.cdist(xz, xz)
neg = d >= 20.0
hit &= neg
print("negatives in marked bins: %d of %d negatives (%.3f %%)" % (int(hit.sum()), int(neg.sum()), 100.0 * float(hit.sum()) / float(neg.sum())))
inr = (mat >= float(thr.min())) & (mat <= float(thr.max()))
print("scores inside [thr min, thr max]: %.2f %%" % (100.0 * float(inr.float().mean())))
R4, S = (r + 3) // 4, (m + 255) // 256
pad = torch.zeros(R4 * 4, S * 256, dtype=torch.bool, device="cuda")
pad[:r, :m] = hit
blk = pad.view(R4, 4, S, 256).any(3).any(1)
print("blocks (4 x 256) with such a negative: %.2f %%" % (100.0 * float(blk.float().mean())))
for w in (9, 10, 11):
    coarse = torch.zeros(NBP + 32 >> w << w + 1 >> w, dtype=torch.bool)
    cb = np.unique(bins >> w)
    print("coarse buckets of %d bins: %d hold a marked bin" % (1 << w, len(cb)), end="; ")
    cm = torch.zeros((NBP >> w) + 2, dtype=torch.bool, device="cuda")
    cm[torch.from_numpy(cb.astype(np.int64)).cuda()] = True
    chit = cm[(key >> w).long()] & neg
    pad[:r, :m] = chit
    print("blocks with a negative in one of them: %.2f %%" % (100.0 * float(pad.view(R4, 4, S, 256).any(3).any(1).float().mean())))
hist = torch.bincount(key[neg].long().flatten(), minlength=NBP).cpu().numpy()
cs = np.cumsum(hist) / hist.sum()
print("negatives' keys: 10 %% below bin %d, 50 %% below %d, 90 %% below %d, 99 %% below %d (bin of 0.5: %d, of 1.0: %d)" % tuple(
    [int(np.searchsorted(cs, q)) for q in (0.1, 0.5, 0.9, 0.99)] + [HALF, ONE]))
