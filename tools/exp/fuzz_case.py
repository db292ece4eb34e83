#!/usr/bin/env python3
"""One shape of tools/exp/fuzz_oracle.py in detail: usage fuzz_case.py <n> <k>  (replays the fuzz's random stream up to that shape)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sg_pr_amd import engine, synth
from oracle import sgpr_oracle as oracle
want_n, want_k = int(sys.argv[1]), int(sys.argv[2])
sd = torch.load("tests/golden/model.pth", map_location="cpu")
osd = oracle.load_checkpoint("tests/golden/model.pth")
eng = engine.Engine(sd)
rng = np.random.default_rng(2024)
for trial in range(40):
    n = int(rng.integers(17, 257))
    k = int(rng.integers(1, min(32, n // 2) + 1))
    hi = n - k
    lo = int(rng.integers(1, hi + 1))
    g = 8
    seed, kl = int(rng.integers(1 << 30)), bool(rng.integers(2))
    if (n, k) != (want_n, want_k):
        continue
    c, l, n_real = synth.make_graphs(g, n, lo, hi, seed, kitti_like=kl)
    print("trial", trial, "n", n, "k", k, "lo", lo, "hi", hi, "n_real", n_real.tolist(), "kitti_like", kl)
    dense = torch.from_numpy(synth.dense_features(c, l))
    rp, ra, remb = oracle.embed(osd, dense, k)
    pooled, att, emb, layers, knn = eng.embed(c, l, k, debug=True)
    ref_layers = oracle.conv_pass(osd, dense, k, want_layers=True)[1]
    print("per graph max|d att|", (att.cpu() - ra.squeeze(-1)).abs().amax(1).tolist())
    print("per graph max|d pooled|", (pooled.cpu() - rp).abs().amax(1).tolist())
    print("per graph max|d emb|", (emb.cpu() - remb).abs().amax((1, 2)).tolist())
    if ref_layers is not None:
        names = ["xyz1", "xyz2", "xyz3", "sem1", "sem2", "sem3"]
        for li, nm in enumerate(names):
            ref = ref_layers[nm].permute(0, 2, 1)
            got = layers[:, li, :, : ref.shape[2]].cpu()
            print(nm, "per graph max|d|", (got - ref).abs().amax((1, 2)).tolist())
    p2, a2, _ = eng.embed(c, l, k, want_att=True)
    print("production vs debug instance: pooled equal", torch.equal(p2, pooled), "max|d|", (p2 - pooled).abs().max().item())
    # which neighbour sets differ, layer by layer, and by what margin?  Judged through the reference's own fp32 keys of
    # the selected nodes (indifferent to the order among equal keys); float64 distances of the layer's reference input
    # give the true margin between the k-th and (k+1)-th candidate.
    order = ["xyz1", "xyz2", "xyz3", "sem1", "sem2", "sem3"]                       # dump order of `knn`
    inputs = {"xyz1": dense[:, :3, :], "sem1": dense[:, 3:, :]}
    if ref_layers is not None:
        inputs.update({"xyz2": ref_layers["xyz1"], "xyz3": ref_layers["xyz2"], "sem2": ref_layers["sem1"], "sem3": ref_layers["sem2"]})
    for li, nm in enumerate(order):
        if nm not in inputs:
            continue
        x = inputs[nm]
        pd = oracle.neg_sq_dist(x)
        kk = knn[:, li].cpu().numpy().astype(np.int64)
        ref_vals = np.sort(pd.topk(k=k, dim=-1)[0].numpy(), -1)
        got_vals = np.sort(np.take_along_axis(pd.numpy(), kk, -1), -1)
        bad = np.argwhere((ref_vals != got_vals).any(-1))
        x64 = x.double().numpy().transpose(0, 2, 1)
        print(nm, "rows with another neighbour set:", len(bad))
        for b, i in bad[:6]:
            d2 = ((x64[b, i][None, :] - x64[b]) ** 2).sum(-1)
            srt = np.sort(d2)
            ridx = pd[b, i].topk(k)[1].numpy()
            only_e, only_r = sorted(set(kk[b, i]) - set(ridx)), sorted(set(ridx) - set(kk[b, i]))
            print("  graph", b, "row", i, "engine-only", only_e, [float(d2[j]) for j in only_e], "reference-only", only_r,
                  [float(d2[j]) for j in only_r], "| exact k-th / (k+1)-th d2: %.9f / %.9f" % (srt[k - 1], srt[k]),
                  "| reference fp32 keys:", [float(pd[b, i, j]) for j in only_e], [float(pd[b, i, j]) for j in only_r])
