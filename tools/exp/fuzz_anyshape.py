#!/usr/bin/env python3
"""Fuzz: the any-shape kernels (sgpr_generic.hip) vs the CPU oracle on random ARCHITECTURES, node_num and K - through the
reference's SG API: embeddings, attention, pooled vectors, all-pairs / list / pair scores, packed = ragged = dense inputs.
Graphs get at least K padding slots (one-hot rows tie exactly across labels otherwise: torch.topk's tie order decides).
  python tools/exp/fuzz_anyshape.py [trials] [wide]   (wide: architectures inside sgpr_wide.hip's matrix-core limits)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from sg_pr_amd import sg_net, synth  # noqa: E402
from sg_pr_amd.allpairs import RaggedGraphs  # noqa: E402
from sg_pr_amd.parser_sg import sgpr_args  # noqa: E402
from oracle import sgpr_oracle as oracle  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 24
wide = len(sys.argv) > 2 and sys.argv[2] == "wide"
rng = np.random.default_rng(77)
worst = {"emb": 0.0, "att": 0.0, "pooled": 0.0, "score": 0.0}
for trial in range(trials):
    labels = int(rng.integers(1, 65))
    f1, f2 = int(rng.integers(1, 257)), int(rng.integers(1, 257))
    f3 = int(rng.integers(1, 129))
    t, bn = int(rng.integers(1, 65)), int(rng.integers(1, 65))
    if trial % 4 == 0:                                             # force the larger-than-built property one way or another
        f3 = int(rng.integers(33, 129))
    elif max(labels - 12, f1 - 64, f2 - 64, f3 - 32, t - 16, bn - 16) <= 0:
        labels = int(rng.integers(13, 65))
    n = int(rng.integers(2, 200)) if trial % 5 else int(rng.integers(257, 700))
    k = int(rng.integers(1, min(64, max(1, n // 2)) + 1))
    if wide:            # inside sgpr_wide.hip's limits: the matrix-core any-shape embed and its dense tail
        labels = int(rng.integers(1, 33))
        f1, f2, f3 = int(rng.integers(1, 129)), int(rng.integers(1, 129)), int(rng.integers(1, 65))
        t, bn = int(rng.integers(1, 33)), int(rng.integers(1, 33))
        if max(labels - 12, f1 - 64, f2 - 64, f3 - 32, t - 16, bn - 16) <= 0:
            f1 = int(rng.integers(65, 129))
        n, k = int(rng.integers(21, 113)), 10
    args = sgpr_args()
    args.filters_1, args.filters_2, args.filters_3, args.tensor_neurons, args.bottle_neck_neurons = f1, f2, f3, t, bn
    args.node_num, args.K = n, k
    torch.manual_seed(trial)
    model = sg_net.SG(args, labels)
    with torch.no_grad():
        for name, buf in model.named_buffers():
            if name.endswith("running_mean"):
                buf.copy_(torch.randn_like(buf) * 0.2)
            if name.endswith("running_var"):
                buf.copy_(torch.rand_like(buf) + 0.5)
        for name, prm in model.named_parameters():
            if name.endswith(".1.weight"):
                prm.copy_(torch.rand_like(prm) + 0.5)
            if name.endswith(".1.bias"):
                prm.copy_(torch.randn_like(prm) * 0.2)
    model.eval()
    sd = {kk: v.detach().clone() for kk, v in model.state_dict().items()}
    eng = model.engine()
    assert eng.any_shape and eng.pw == f3
    g = 6
    hi = max(1, n - k)
    lo = max(1, hi // 3)
    c, l, _ = synth.make_graphs(g, n, lo, hi, int(rng.integers(1 << 30)))
    l = np.where(l >= 0, (l * 5 + 3 + np.arange(g)[:, None]) % labels, l).astype(np.int32)
    for gi in range(g):
        m = int((l[gi] >= 0).sum())
        order = np.argsort(l[gi, :m], kind="stable")
        l[gi, :m], c[gi, :m] = l[gi, :m][order], c[gi, :m][order]
    feats = torch.from_numpy(synth.dense_features(c, l, num_labels=labels))
    ref_emb = oracle.conv_pass(sd, feats, k)
    ref_pooled, ref_att = oracle.embed(sd, feats, k)[:2]
    pooled, att, emb = model.embed(c, l, want_att=True, want_emb=True)
    scale = max(1.0, float(ref_emb.abs().max()))
    d_emb = (emb.cpu() - ref_emb).abs().amax(dim=(1, 2)) / scale
    ok = d_emb < 2e-5                                              # graphs without a flipped near-tie
    tag = (trial, labels, f1, f2, f3, t, bn, n, k)
    assert int(ok.sum()) >= g - 2, (tag, d_emb)
    d_att = float((att.cpu() - ref_att.reshape(g, n))[ok].abs().max())
    d_pool = float(((pooled.cpu() - ref_pooled)[ok].abs().max()) / max(1.0, float(ref_pooled.abs().max())))
    assert d_att < 1e-4 and d_pool < 2e-4, (tag, d_att, d_pool)
    rag = RaggedGraphs.from_padded(c, l, device="cuda", num_labels=labels)
    assert torch.equal(model.embed(rag, None)[0], pooled), tag
    assert torch.equal(eng.embed_dense(feats, k)[0], pooled), tag
    ours = pooled.cpu()
    mat = model.score_all_pairs(pooled, pooled).cpu()
    want = oracle.score_all_pairs(sd, ours, ours)                   # the tail on equal inputs
    d_s = float((mat - want).abs().max())
    i1 = rng.integers(0, g, 50).astype(np.int32)
    i2 = rng.integers(0, g, 50).astype(np.int32)
    lst = model.score_pooled(pooled, pooled, torch.from_numpy(i1), torch.from_numpy(i2)).cpu()
    d_l = float((lst - want[torch.from_numpy(i1).long(), torch.from_numpy(i2).long()]).abs().max())
    got, a1, a2 = model({"features_1": feats[:3], "features_2": feats[3:]})
    d_f = float((got.cpu() - want[torch.arange(3), torch.arange(3, 6)]).abs().max())
    assert max(d_s, d_l, d_f) < 1e-4, (tag, d_s, d_l, d_f)
    eng.check_status()
    worst["emb"] = max(worst["emb"], float(d_emb[ok].max()))
    worst["att"] = max(worst["att"], d_att)
    worst["pooled"] = max(worst["pooled"], d_pool)
    worst["score"] = max(worst["score"], d_s, d_l, d_f)
    print("trial %2d labels %2d filters %3d/%3d/%3d neurons %2d/%2d node_num %3d K %2d: %d/%d graphs at rounding level, "
          "emb %.1e att %.1e pooled %.1e score %.1e" % (trial, labels, f1, f2, f3, t, bn, n, k, int(ok.sum()), g,
                                                       float(d_emb[ok].max()), d_att, d_pool, max(d_s, d_l, d_f)))
print("fuzz ok: %d architectures / shapes, worst (relative) emb %.1e, att %.1e, pooled %.1e, score on equal inputs %.1e"
      % (trials, worst["emb"], worst["att"], worst["pooled"], worst["score"]))
