#!/usr/bin/env python3
"""Start / end time and place (XCD, SE, CU) of every workgroup of the KITTI-00 embed launch (a library built with
-DSGPR_EMBED_STAMPS=1: tools/build_variant.sh estamps -DSGPR_EMBED_STAMPS=1; SGPR_HIP_LIB=variants/libsgpr_estamps.so).
How long is a workgroup, how long does a freed slot wait for the next one, how ragged is the end of the launch?"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sg_pr_amd import engine, synth
sd = torch.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "model.pth"), map_location="cpu")
eng = engine.Engine(sd)
stress = len(sys.argv) > 1 and sys.argv[1] == "stress"      # BASELINE config 5 (embed_big_kernel: one workgroup per CU)
if stress:
    c, l, _ = synth.config5_pairs(seed=0)
    g, K, SLOTS = c.shape[0], 20, 1
else:
    g, K, SLOTS = 4541, 10, 4
    c, l, _, _ = synth.kitti_like_sequence(g, 100, 0)
order, cap = eng.size_order(c, l, K)
cd, ld = torch.from_numpy(c).cuda(), torch.from_numpy(l).cuda()
for _ in range(30):
    p = eng.embed(cd, ld, K, node_cap=cap, order=order)[0]
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    p = eng.embed(cd, ld, K, node_cap=cap, order=order)[0]
e1.record()
torch.cuda.synchronize()
print("embed call %.1f us (events, 20 calls; stamps cost two barriers + three stores per workgroup)" % (e0.elapsed_time(e1) / 20 * 1e3))
out = np.zeros(32768 * 4, dtype=np.uint64)
eng.lib.sgpr_debug_embed_stamps.argtypes = [ctypes.c_void_p]
assert eng.lib.sgpr_debug_embed_stamps(out.ctypes.data) == 0
st = out.reshape(32768, 4)[:g]
t0 = st[:, 0].astype(np.int64).min()
start = (st[:, 0].astype(np.int64) - t0) / 100.0
end = (st[:, 2].astype(np.int64) - t0) / 100.0
hw = (st[:, 1] & np.uint64(0xffffffff)).astype(np.int64)
xcc = (st[:, 1] >> np.uint64(32)).astype(np.int64) & 0xf
cu = (hw >> 8) & 0xf
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
place = ((xcc * 8 + se) * 2 + sh) * 16 + cu
dur = end - start
n = (l >= 0).sum(1)[order.cpu().numpy() if hasattr(order, "cpu") else order]
print("launch: first start 0.0, last end %.1f us; workgroup duration mean %.2f median %.2f p90 %.2f max %.2f us" % (
    end.max(), dur.mean(), np.median(dur), np.percentile(dur, 90), dur.max()))
print("sum of durations / (CUs x SLOTS slots) = %.1f us of %.1f us: %.1f %% of the slot time is inside a workgroup" % (
    dur.sum() / (eng.num_cus * SLOTS), end.max(), 100.0 * dur.sum() / (eng.num_cus * SLOTS) / end.max()))
places = np.unique(place)
per_place = np.unique(place, return_counts=True)[1]
print("distinct places (XCD, SE, SH, CU): %d; workgroups per place min %d max %d" % (len(places), per_place.min(), per_place.max()))
gaps, busy_end = [], []
for pl in places:
    idx = np.nonzero(place == pl)[0]
    s_, e_ = start[idx], end[idx]
    o = np.argsort(s_)
    s_, e_ = s_[o], e_[o]
    # four slots per CU: greedy assignment of each workgroup to the slot that freed last before its start
    free = []
    for a, b in zip(s_, e_):
        cand = [f for f in free if f <= a + 0.05]
        if cand:
            f = max(cand)
            free.remove(f)
            gaps.append(a - f)
        free.append(b)
    busy_end.append(e_.max())
gaps = np.array(gaps)
print("slot refill gap (end of a workgroup -> start of the next on that CU): mean %.2f median %.2f p90 %.2f us over %d refills" % (
    gaps.mean(), np.median(gaps), np.percentile(gaps, 90), len(gaps)))
be = np.array(busy_end)
print("last end per CU: min %.1f median %.1f max %.1f us (ragged end: %.1f us)" % (be.min(), np.median(be), be.max(), be.max() - be.min()))
for lo in ((0, 100, 200, 300, 400, 480) if stress else (0, 25, 50, 75, 100, 120)):
    m = (start >= lo) & (start < lo + (20 if stress else 5))
    if m.any():
        print("  workgroups starting in [%3d, %3d) us: %4d, mean duration %.2f us, mean slots of their graphs %.1f" % (lo, lo + 5, int(m.sum()), dur[m].mean(), n[m].mean()))
