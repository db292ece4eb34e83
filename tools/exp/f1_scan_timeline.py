#!/usr/bin/env python3
"""Per-wave time stamps of sgpr_f1_max's pass A (a library built with -DSGPR_F1_SCAN_STAMPS=1: tools/build_variant.sh stamps
-DSGPR_F1_SCAN_STAMPS=1; SGPR_HIP_LIB=variants/libsgpr_stamps.so).  100 MHz counter."""
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
for _ in range(20):
    rc = lib.sgpr_f1_max(h, ctypes.c_void_p(mat.data_ptr()), r, m, mat.stride(0), 0, ctypes.c_void_p(xz.data_ptr()), 3.0, 20.0, None, m,
                         ctypes.c_void_p(res.data_ptr()), ctypes.c_void_p(ws.data_ptr()), ws_bytes, stream)
    assert rc == 0
torch.cuda.synchronize()
out = np.zeros(8192 * 8, dtype=np.uint64)
lib.sgpr_debug_f1_scan_stamps.argtypes = [ctypes.c_void_p]
assert lib.sgpr_debug_f1_scan_stamps(out.ctypes.data) == 0
st = out.reshape(8192, 8)[:eng.num_cus * 16].astype(np.int64)
t0 = st[:, 0].min()
us = (st - t0) / 100.0
names = ["start", "zeroed", "streamed", "classified", "flushed", "barrier", "slab written"]
for i, n in enumerate(names):
    col = us[:, i]
    print("%-13s min %6.1f  median %6.1f  p90 %6.1f  max %6.1f us" % (n, col.min(), np.median(col), np.percentile(col, 90), col.max()))
d = us[:, 3] - us[:, 2]
print("stream end -> classified (barrier + the shared list) per wave: median %.1f, p90 %.1f, max %.1f us; waves with one: %d of %d" % (np.median(d), np.percentile(d, 90), d.max(), int((d > 0.3).sum()), len(d)))
d = us[:, 2] - us[:, 1]
print("streaming phase per wave: median %.1f, p90 %.1f, max %.1f us" % (np.median(d), np.percentile(d, 90), d.max()))
