"""Pin the CPU oracle against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_checkpoint_layout(oracle_sd):
    # SURVEY.md §8 row a-ckpt: 50 keys, 48 696 scalars
    assert len(oracle_sd) == 50
    assert sum(v.numel() for v in oracle_sd.values()) == 48696
    assert oracle_sd["tensor_network.weight_matrix"].shape == (32, 32, 16)
    assert oracle_sd["dgcnn_f_conv1.0.weight"].shape == (64, 24, 1, 1)


def test_pack_matches_reference(oracle, golden_dir):
    g = _load(golden_dir, "kitti3_n100_k10.npz")
    for i, n in enumerate(g["names"]):
        d = oracle.read_graph(os.path.join(golden_dir, "data", str(n) + ".json"))
        f = oracle.pack_graph(d["centers"], d["nodes"], 100)
        np.testing.assert_array_equal(f.astype(np.float32), g["features"][i])


def test_process_pair_distance(oracle, golden_dir):
    g = _load(golden_dir, "kitti3_n100_k10.npz")
    names = [str(n) for n in g["names"]]
    for (i, j), dist in zip(g["pair_ij"], g["distance"]):
        d = oracle.process_pair([os.path.join(golden_dir, "data", names[i] + ".json"),
                                 os.path.join(golden_dir, "data", names[j] + ".json")])
        assert d["distance"] == dist
    assert abs(g["distance"][2] - 133.12761323772054) < 1e-9


def test_intermediates_shipped_graphs(oracle, oracle_sd, golden_dir):
    g = _load(golden_dir, "kitti3_n100_k10.npz")
    feats = torch.from_numpy(g["features"])
    with torch.no_grad():
        e, layers = oracle.conv_pass(oracle_sd, feats, 10, want_layers=True)
        p, a = oracle.attention(oracle_sd, e)
    for name, val in layers.items():
        np.testing.assert_allclose(val.numpy(), g[name], rtol=0, atol=2e-6, err_msg=name)
    np.testing.assert_allclose(e.numpy(), g["emb"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(p.numpy().reshape(3, -1), g["pooled"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(a.numpy().reshape(3, -1), g["att"], rtol=0, atol=1e-6)
    # known answers quoted in SURVEY.md §4
    np.testing.assert_allclose(np.linalg.norm(g["pooled"], axis=1), [13.553494, 13.984426, 14.363490], atol=2e-5)
    # kNN neighbour SETS of the first (xyz) layer agree with the reference
    idx = oracle.knn(feats[:, :3, :], 10).numpy()
    assert np.array_equal(np.sort(idx, -1), np.sort(g["knn_idx"][:, 0].astype(np.int64), -1))


def test_scores_nine_pairs(oracle, oracle_sd, golden_dir):
    g = _load(golden_dir, "kitti3_n100_k10.npz")
    feats = torch.from_numpy(g["features"])
    f1 = torch.stack([feats[i] for i, _ in g["pair_ij"]])
    f2 = torch.stack([feats[j] for _, j in g["pair_ij"]])
    s, a1, a2 = oracle.forward(oracle_sd, f1, f2, 10)
    np.testing.assert_allclose(s.numpy(), g["scores"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(s.numpy(), g["scores_batched"], rtol=0, atol=1e-6)
    # BASELINE.md §2 table
    assert abs(s[2].item() - 1.3489922e-06) < 1e-9
    assert abs(s[0].item() - 0.99934918) < 1e-6
    # NTN vector and the pooled-only path
    p = torch.from_numpy(g["pooled"])
    pi = torch.stack([p[i] for i, _ in g["pair_ij"]])
    pj = torch.stack([p[j] for _, j in g["pair_ij"]])
    with torch.no_grad():
        t = oracle.tensor_network(oracle_sd, pi.unsqueeze(-1), pj.unsqueeze(-1))
    np.testing.assert_allclose(t.numpy().reshape(9, -1), g["ntn"], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(oracle.score_from_pooled(oracle_sd, pi, pj).numpy(), g["scores"], atol=1e-6)
    m = oracle.score_all_pairs(oracle_sd, p, p)
    np.testing.assert_allclose(m.numpy().reshape(-1), g["scores"], atol=1e-6)


def test_eval_batch_pair(oracle, oracle_sd, golden_dir):
    g = _load(golden_dir, "kitti3_n100_k10.npz")
    names = [str(n) for n in g["names"]]
    batch = [[os.path.join(golden_dir, "data", names[i] + ".json"),
              os.path.join(golden_dir, "data", names[j] + ".json")] for i, j in g["pair_ij"]]
    pred, gt = oracle.eval_batch_pair(oracle_sd, batch, 100, 10, 3)
    assert pred.dtype == np.float32 and gt.dtype == np.float64
    np.testing.assert_allclose(pred, g["eval_batch_pred"], atol=1e-6)
    np.testing.assert_array_equal(gt, g["eval_batch_gt"])


@pytest.mark.parametrize("fname", ["synth_n64_k10.npz", "synth_n100_k10.npz", "synth_n256_k20.npz"])
def test_synthetic(oracle, oracle_sd, golden_dir, fname):
    from sg_pr_amd import synth
    g = _load(golden_dir, fname)
    k = int(g["k"])
    dense = torch.from_numpy(synth.dense_features(g["centers"], g["labels"]))
    s, _, _ = oracle.forward(oracle_sd, dense[0::2], dense[1::2], k)
    np.testing.assert_allclose(s.numpy(), g["scores"], rtol=0, atol=2e-6)
    p, a, e = oracle.embed(oracle_sd, dense, k)
    np.testing.assert_allclose(p.numpy(), g["pooled"], rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(a.numpy(), g["att"], rtol=0, atol=2e-6)
    # generator is reproducible from its seed
    c2, l2, _ = synth.make_graphs(g["centers"].shape[0], int(g["node_num"]),
                                  int(g["n_real"].min()), int(g["n_real"].max()), int(g["seed"]))
    assert c2.shape == g["centers"].shape and l2.dtype == np.int32


def test_synth_generator_reproduces_golden_inputs(golden_dir):
    from sg_pr_amd import synth
    g = _load(golden_dir, "synth_n64_k10.npz")
    c, l, n = synth.make_graphs(32, 64, 20, 54, 0)
    np.testing.assert_array_equal(c, g["centers"])
    np.testing.assert_array_equal(l, g["labels"])
    assert (n <= 64 - 10).all()


def test_f1_max(oracle, golden_dir):
    g = _load(golden_dir, "prf1.npz")
    for c in range(int(g["ncases"])):
        p, r, _ = oracle.precision_recall_curve(g[f"gt{c}"], g[f"score{c}"])
        np.testing.assert_allclose(p, g[f"precision{c}"], atol=1e-12)
        np.testing.assert_allclose(r, g[f"recall{c}"], atol=1e-12)
        assert abs(oracle.f1_max(g[f"gt{c}"], g[f"score{c}"]) - float(g[f"f1max{c}"])) < 1e-12


def test_f1_max_vs_sklearn(oracle):
    metrics = pytest.importorskip("sklearn.metrics")
    rng = np.random.default_rng(11)
    gt = (rng.random(777) < 0.1).astype(np.float64)
    sc = rng.random(777).astype(np.float32)
    p, r, _ = metrics.precision_recall_curve(gt, sc)
    with np.errstate(divide="ignore", invalid="ignore"):
        f1 = np.nan_to_num(2 * p * r / (p + r))
    assert abs(oracle.f1_max(gt, sc) - f1.max()) < 1e-12


def test_edges_of_the_tie_regime(oracle, oracle_sd, golden_dir):
    """k-1 / k padded slots, one label, < 17 nodes, no padding with >= k nodes per label, trailing duplicate real
    nodes (every tie is between feature-identical nodes, so the reference itself is deterministic)."""
    from sg_pr_amd import synth
    g = _load(golden_dir, "edge_n100_k10.npz")
    dense = torch.from_numpy(synth.dense_features(g["centers"], g["labels"]))
    pooled, att, emb = oracle.embed(oracle_sd, dense, 10)
    np.testing.assert_allclose(emb.numpy(), g["emb"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(att.numpy().reshape(12, -1), g["att"], rtol=0, atol=2e-6)
    s = oracle.score_all_pairs(oracle_sd, pooled, pooled).numpy()
    np.testing.assert_allclose(s, g["scores"], rtol=0, atol=5e-6)


def test_all_release_checkpoints(oracle, release_state_dicts, golden_dir):
    """The 18 checkpoints of model/release_model.zip: different BatchNorm statistics and weights through the same
    restatement - the nine shipped pairs and a synthetic batch under each."""
    from sg_pr_amd import synth
    g = _load(golden_dir, "release_models.npz")
    k3 = _load(golden_dir, "kitti3_n100_k10.npz")
    feats = torch.from_numpy(k3["features"])
    i, j = g["pair_ij"][:, 0].astype(np.int64), g["pair_ij"][:, 1].astype(np.int64)
    dsyn = torch.from_numpy(synth.dense_features(g["syn_centers"], g["syn_labels"]))
    assert sorted(release_state_dicts) == sorted(str(n) for n in g["names"]) and len(release_state_dicts) == 18
    for name, sd in release_state_dicts.items():
        assert len(sd) == 50                                                    # strict layout (SURVEY.md a-ckpt)
        s9, _, _ = oracle.forward(sd, feats[i], feats[j], 10)
        np.testing.assert_allclose(s9.numpy(), g["scores9/" + name], rtol=0, atol=5e-6, err_msg=name)
        ss, _, _ = oracle.forward(sd, dsyn[0::2], dsyn[1::2], 10)
        np.testing.assert_allclose(ss.numpy(), g["scores_syn/" + name], rtol=0, atol=5e-6, err_msg=name)


def test_coordinate_keys_operation_order(oracle):
    """The HIP path restates the reference's fp32 arithmetic of the first xyz layer's ranking keys operation for
    operation (sgpr_embed.hip gram_xyz_direct): this pins what that arithmetic IS as torch evaluates dgcnn.py:15-17 -
    the K=3 matmul as the FMA chain over x, y, z, |x|^2 = (x*x + y*y) + z*z with every step rounded, then
    (-|x_j|^2 - inner) - |x_i|^2.  (FMA emulated in float64: products of two fp32 are exact there.)"""
    rng = np.random.default_rng(5)
    f32, f64 = np.float32, np.float64

    def fma(a, b, c):
        return (a.astype(f64) * b.astype(f64) + c.astype(f64)).astype(f32)

    for n in (44, 100, 135):
        x = rng.uniform(-50, 50, (4, 3, n)).astype(f32)
        pd = oracle.neg_sq_dist(torch.from_numpy(x)).numpy()
        xi, xj = x[:, :, :, None], x[:, :, None, :]
        dot = fma(xi[:, 2], xj[:, 2], fma(xi[:, 1], xj[:, 1], (xi[:, 0] * xj[:, 0]).astype(f32)))
        sq = (x * x).astype(f32)
        n2 = ((sq[:, 0] + sq[:, 1]).astype(f32) + sq[:, 2]).astype(f32)
        t = fma(np.full_like(dot, 2), dot, -n2[:, None, :])
        mine = (t - n2[:, :, None]).astype(f32)
        # the float64 emulation double-rounds about one FMA in 1e7; everything else must be bit-identical
        assert np.mean(mine != pd) < 1e-5
