#!/usr/bin/env python3
"""Where does sgpr_f1_max spend its time?  Wall time per call after a proper warm-up, and the phase stamps of the
single-workgroup plan kernel (thread 0, 100 MHz real-time counter, workspace bytes 128..176):
positives by bin | scans, marks | hash de-duplication | sort | per-threshold info.
usage: f1_phases.py [world|kitti] [reps]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sg_pr_amd import allpairs, engine, synth
kind = sys.argv[1] if len(sys.argv) > 1 else "kitti"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
sd = torch.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "model.pth"), map_location="cpu")
eng = engine.Engine(sd)
gen = synth.world_sequence if kind == "world" else synth.kitti_like_sequence
c, l, _, poses = gen(4541, 100, seed=0)
order, cap = eng.size_order(c, l, 10)
p = eng.embed(torch.from_numpy(c).cuda(), torch.from_numpy(l).cuda(), 10, node_cap=cap, order=order)[0]
mat = eng.score_all_pairs(p, p)
xz = allpairs.pose_xz(poses).cuda()
t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.5:                  # clock ramp
    eng.score_all_pairs(p, p, out=mat)
torch.cuda.synchronize()
lib, h = eng.lib, eng._h
r, m = mat.shape
ws_bytes = lib.sgpr_f1_max_workspace_bytes(h, r, m)
ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
res = torch.empty(8, dtype=torch.float64, device="cuda")
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def call():
    rc = lib.sgpr_f1_max(h, ctypes.c_void_p(mat.data_ptr()), r, m, mat.stride(0), 0, ctypes.c_void_p(xz.data_ptr()), 3.0, 20.0, None, m,
                         ctypes.c_void_p(res.data_ptr()), ctypes.c_void_p(ws.data_ptr()), ws_bytes, stream)
    assert rc == 0
for _ in range(5):
    call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    call()
e1.record()
torch.cuda.synchronize()
print("%s sequence: sgpr_f1_max %.1f us per call (events around %d back-to-back calls); result %s" % (kind, e0.elapsed_time(e1) / reps * 1e3, reps, res.cpu().tolist()))
st = ws[128:176].cpu().numpy().view(np.uint64).astype(np.int64)
names = ["positives by bin", "scans + marks", "hash de-duplication", "compaction + sort", "per-threshold info"]
print("plan kernel phases (us): " + ", ".join("%s %.1f" % (n, (st[i + 1] - st[i]) / 100.0) for i, n in enumerate(names) if st[i + 1] >= st[i] > 0))
t0 = time.perf_counter()
for _ in range(reps):
    call()
    out = res.cpu()
print("with the 64-byte copy back and a synchronisation per call: %.1f us" % ((time.perf_counter() - t0) / reps * 1e6))
