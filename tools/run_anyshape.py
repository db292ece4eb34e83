#!/usr/bin/env python3
"""What the any-shape kernels (sgpr_generic.hip: plain fp32, one 256-thread workgroup per graph, activations in global scratch)
cost next to the tuned ones: launch times by HIP events around back-to-back launches.
  python tools/run_anyshape.py [out.txt]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from sg_pr_amd import engine, sg_net, synth  # noqa: E402
from sg_pr_amd.parser_sg import sgpr_args  # noqa: E402

lines = []


def log(s):
    print(s)
    lines.append(s)


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3          # us


sd = torch.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "model.pth"), map_location="cpu")
sd = {k[7:] if k.startswith("module.") else k: v for k, v in sd.items()}
tuned = engine.Engine(sd)
# the shipped checkpoint as a 13-label model (one unused label channel with zero weights): the same function on an any-shape handle
sd13 = {k: v.clone() for k, v in sd.items()}
w = sd13["dgcnn_f_conv1.0.weight"]
sd13["dgcnn_f_conv1.0.weight"] = torch.cat((w.reshape(w.shape[0], 2, 12), torch.zeros(w.shape[0], 2, 1)), dim=2).reshape(w.shape[0], 26, 1, 1)
any13 = engine.Engine(sd13, engine.SgprDims(13, 64, 64, 32, 16, 16))
assert any13.any_shape and not tuned.any_shape

log("# python tools/run_anyshape.py  (us per launch, HIP events around 10 back-to-back launches)")
c, l, _, _ = synth.kitti_like_sequence(4541, 100, 0)
order, cap = tuned.size_order(c, l, 10)
cd, ld = torch.from_numpy(c).cuda(), torch.from_numpy(l).cuda()
t_t = timed(lambda: tuned.embed(cd, ld, 10, node_cap=cap, order=order))
# (debug bit 23: the plain-fp32 any-shape kernel; without it a handle inside sgpr_wide.hip's limits embeds on the matrix cores)
any13.set_skip_mask(1 << 23)
t_a = timed(lambda: any13.embed(cd, ld, 10), reps=3)
p_t = tuned.embed(cd, ld, 10, node_cap=cap, order=order)[0]
p_a = any13.embed(cd, ld, 10)[0]
any13.set_skip_mask(0)
t_w = timed(lambda: any13.embed(cd, ld, 10), reps=3)
p_w = any13.embed(cd, ld, 10)[0]
dev_w = (p_w - p_a).abs().amax(1)
log("... the same handle on the matrix-core any-shape embed (sgpr_wide.hip; every slot processed, no super-nodes): %.1f us (%.1f x "
    "the tuned kernel, %.1f x faster than plain fp32); |d pooled| against the plain-fp32 kernel: median %.1e, %d graphs above 2e-4, max %.2e"
    % (t_w, t_w / t_t, t_a / t_w, float(dev_w.median()), int((dev_w > 2e-4).sum()), float(dev_w.max())))
dev = (p_t - p_a).abs().amax(1)
log("KITTI-00 shape (4541 graphs, node_num 100, K 10), shipped weights: embed (both launches of the call) tuned %.1f us, "
    "any-shape %.1f us (%.0f x); |d pooled| between them: median %.1e, %d graphs above 2e-4 (the near-tied neighbours of "
    "profiles/*_seq_parity.txt: 24-bit against fp32 keys), max %.2e"
    % (t_t, t_a, t_a / t_t, float(dev.median()), int((dev > 2e-4).sum()), float(dev.max())))
s_t = timed(lambda: tuned.score_all_pairs(p_t, p_t))
any13.set_skip_mask(1 << 23)
s_a = timed(lambda: any13.score_all_pairs(p_t, p_t), reps=3)
m_t, m_a = tuned.score_all_pairs(p_t, p_t), any13.score_all_pairs(p_t, p_t)
any13.set_skip_mask(0)
s_w = timed(lambda: any13.score_all_pairs(p_t, p_t), reps=3)
m_w = any13.score_all_pairs(p_t, p_t)
log("all-pairs tail 4541 x 4541: tuned %.1f us (%.1f G pairs/s), any-shape plain fp32 %.1f us (%.2f G pairs/s, %.0f x); max |d score| %.2e"
    % (s_t, 4541 ** 2 / s_t / 1e3, s_a, 4541 ** 2 / s_a / 1e3, s_a / s_t, float((m_t - m_a).abs().max())))
log("... the same handle's tail on the matrix cores (sgpr_wide.hip, three launches: prep, tail, the gated plain kernel): %.1f us "
    "(%.2f G pairs/s, %.1f x the tuned kernel, %.1f x faster than plain fp32); max |d score| against plain fp32 %.2e"
    % (s_w, 4541 ** 2 / s_w / 1e3, s_w / s_t, s_a / s_w, float((m_w - m_a).abs().max())))
del m_t, m_a, m_w

# beyond the tuned kernels' node_num / K on the shipped checkpoint
for n, k, g in ((512, 20, 1024), (1024, 10, 512), (100, 40, 1024)):
    cc, ll, _ = synth.make_graphs(g, n, n // 2, n - max(k, 10), seed=n + k)
    ccd, lld = torch.from_numpy(cc).cuda(), torch.from_numpy(ll).cuda()
    t = timed(lambda: tuned.embed(ccd, lld, k), reps=3)
    log("node_num %d, K %d, %d graphs (shipped checkpoint, any-shape embed kernel): %.1f us = %.2f us per graph"
        % (n, k, g, t, t / g))

# a larger architecture through the reference's SG API
for labels, f1, f2, f3, tn, bn in ((12, 128, 128, 64, 32, 32), (30, 256, 256, 128, 64, 64)):
    args = sgpr_args()
    args.filters_1, args.filters_2, args.filters_3, args.tensor_neurons, args.bottle_neck_neurons = f1, f2, f3, tn, bn
    args.node_num, args.K = 100, 10
    torch.manual_seed(1)
    model = sg_net.SG(args, labels).eval()
    eng = model.engine()
    lab = np.where(l >= 0, l % labels, l).astype(np.int32)[:1024]
    cg, lg = torch.from_numpy(c[:1024]).cuda(), torch.from_numpy(lab).cuda()
    t = timed(lambda: eng.embed(cg, lg, 10), reps=3)
    p = eng.embed(cg, lg, 10)[0]
    eng.set_skip_mask(1 << 23)
    t_plain = timed(lambda: eng.embed(cg, lg, 10), reps=3)
    p_plain = eng.embed(cg, lg, 10)[0]
    eng.set_skip_mask(0)
    dv = (p - p_plain).abs().amax(1) / p_plain.abs().amax().clamp(min=1.0)
    s = timed(lambda: eng.score_all_pairs(p, p), reps=3)
    m_w = eng.score_all_pairs(p, p)
    eng.set_skip_mask(1 << 23)
    s_plain = timed(lambda: eng.score_all_pairs(p, p), reps=3)
    d_tail = float((m_w - eng.score_all_pairs(p, p)).abs().max())
    eng.set_skip_mask(0)
    log("architecture {%d labels, filters %d/%d/%d, %d tensor / %d bottleneck neurons}, 1024 graphs of node_num 100: embed %.1f us "
        "(%.2f us per graph; plain fp32 only: %.1f us = %.2f us per graph; relative |d pooled| between the two: median %.1e, max %.1e), "
        "all-pairs 1024 x 1024 %.1f us (%.3f G pairs/s; plain fp32 only: %.1f us; max |d score| between the two %.1e)"
        % (labels, f1, f2, f3, tn, bn, t, t / 1024, t_plain, t_plain / 1024, float(dv.median()), float(dv.max()), s,
           1024 ** 2 / s / 1e3, s_plain, d_tail))
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("\n".join(lines) + "\n")
