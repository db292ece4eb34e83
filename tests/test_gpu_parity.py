"""Parity of the HIP path (through the C-ABI) against the CPU oracle and the
reference-generated golden vectors.  Needs a real MI355X: `pytest -m gpu`.

Bar (BASELINE.json north_star): similarity scores within 1e-4 of the reference.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SCORE_TOL = 1e-4   # north_star: "Similarity scores match the reference within 1e-4"
# Gates below sit just above what the kernels deliver today (tools/diag_tolerances.py prints the observed values: layer
# outputs <= 7.2e-7, embeddings 2.4e-6, attention 7.3e-7, pooled 1.2e-5 (1.2e-4 at node_num 256), golden scores
# 1.4e-6), so that a numerical regression shows long before the 1e-4 bar.
FEAT_TOL = 5e-6    # EdgeConv layer outputs (values up to ~10): folded BN + reassociated dot products
EMB_TOL = 2e-5     # conv_end output
ATT_TOL = 5e-6
GOLDEN_SCORE_TOL = 1e-5


@pytest.fixture(scope="module")
def eng(ckpt_path):
    from sg_pr_amd import engine
    from oracle import sgpr_oracle
    e = engine.Engine(sgpr_oracle.load_checkpoint(ckpt_path), device=0)
    yield e
    e.close()


def _packed_from_golden_features(feats):
    """dense [G,15,N] -> (centers [G,N,3], labels [G,N])."""
    centers = np.ascontiguousarray(feats[:, :3, :].transpose(0, 2, 1))
    onehot = feats[:, 3:, :]
    labels = np.where(onehot.sum(1) > 0, onehot.argmax(1), -1).astype(np.int32)
    return centers, labels


def test_shipped_graphs_every_intermediate(eng, golden_dir):
    g = np.load(os.path.join(golden_dir, "kitti3_n100_k10.npz"))
    centers, labels = _packed_from_golden_features(g["features"])
    pooled, att, emb, layers, knn = eng.embed(centers, labels, 10, debug=True)
    torch.cuda.synchronize()
    layers, knn = layers.cpu().numpy(), knn.cpu().numpy()
    names = ["xyz1", "xyz2", "xyz3", "sem1", "sem2", "sem3"]
    # layer inputs (golden), used to canonicalise neighbour indices: kNN ties are only ever
    # between feature-identical nodes (padding), any of which is an equally valid neighbour
    f = g["features"]
    inputs = [f[:, :3, :], g["xyz1"], g["xyz2"], f[:, 3:, :], g["sem1"], g["sem2"]]
    agree = []
    for li, name in enumerate(names):
        ref = g[name].transpose(0, 2, 1)                       # [G, N, C]
        got = layers[:, li, :, : ref.shape[2]]
        np.testing.assert_allclose(got, ref, rtol=0, atol=FEAT_TOL, err_msg=name)
        x = inputs[li].transpose(0, 2, 1)                      # [G, N, Cin]
        rows_ok = []
        for b in range(x.shape[0]):
            canon = np.array([np.flatnonzero((x[b] == x[b, j]).all(-1))[0] for j in range(x.shape[1])])
            mine = np.sort(canon[knn[b, li]], -1)
            theirs = np.sort(canon[g["knn_idx"][b, li].astype(np.int64)], -1)
            rows_ok.append((mine == theirs).all(-1))
        agree.append(np.mean(rows_ok))
    print("neighbour-set agreement per layer (xyz1..3, sem1..3):", np.round(agree, 4))
    assert (knn >= 0).all() and (knn < 100).all()
    assert min(agree) == 1.0                                    # every neighbour set of every row of every layer
    np.testing.assert_allclose(emb.cpu().numpy(), g["emb"], rtol=0, atol=EMB_TOL)
    np.testing.assert_allclose(att.cpu().numpy(), g["att"], rtol=0, atol=ATT_TOL)
    np.testing.assert_allclose(pooled.cpu().numpy(), g["pooled"], rtol=0, atol=5e-5)
    # nine ordered pairs: pair-list kernel and dense all-pairs kernel
    i1 = torch.tensor(g["pair_ij"][:, 0].astype(np.int32))
    i2 = torch.tensor(g["pair_ij"][:, 1].astype(np.int32))
    s = eng.score_pairs(pooled, pooled, i1, i2).cpu().numpy()
    np.testing.assert_allclose(s, g["scores"], rtol=0, atol=GOLDEN_SCORE_TOL)
    m = eng.score_all_pairs(pooled, pooled).cpu().numpy()
    np.testing.assert_allclose(m.reshape(-1), g["scores"], rtol=0, atol=GOLDEN_SCORE_TOL)
    np.testing.assert_allclose(m.reshape(-1), s, rtol=0, atol=2e-6)
    assert abs(m[0, 2] - 1.3489922e-06) < 1e-6 and abs(m[2, 0] - 2.8918e-05) < 1e-6   # asymmetric NTN


@pytest.mark.parametrize("fname", ["synth_n64_k10.npz", "synth_n100_k10.npz", "synth_n256_k20.npz"])
def test_synthetic_golden(eng, golden_dir, fname):
    g = np.load(os.path.join(golden_dir, fname))
    k = int(g["k"])
    pooled, att, _ = eng.embed(g["centers"], g["labels"], k, want_att=True)
    np.testing.assert_allclose(att.cpu().numpy(), g["att"], rtol=0, atol=ATT_TOL)
    # pooled = a sum over up to 256 nodes of values up to ~25: the gate scales with node_num
    np.testing.assert_allclose(pooled.cpu().numpy(), g["pooled"], rtol=0, atol=3e-4 if int(g["node_num"]) > 128 else 5e-5)
    s = eng.score_pairs(pooled[0::2].contiguous(), pooled[1::2].contiguous()).cpu().numpy()
    np.testing.assert_allclose(s, g["scores"], rtol=0, atol=GOLDEN_SCORE_TOL)


def test_forward_dense_matches_oracle_config2_full(eng, oracle, oracle_sd):
    """BASELINE config 2 at full size: 128 pairs, N=64, k=10 through the drop-in forward."""
    from sg_pr_amd import synth
    centers, labels, _ = synth.config2_pairs(seed=3)
    dense = torch.from_numpy(synth.dense_features(centers, labels))
    f1, f2 = dense[0::2].contiguous(), dense[1::2].contiguous()
    score, a1, a2 = eng.forward_dense(f1, f2, 10)
    ref, r1, r2 = oracle.forward(oracle_sd, f1, f2, 10)
    err = (score.cpu() - ref).abs().max().item()
    print("config2 max|dscore| =", err)
    assert err <= SCORE_TOL
    np.testing.assert_allclose(a1.cpu().numpy(), r1.numpy()[..., 0], rtol=0, atol=ATT_TOL)
    np.testing.assert_allclose(a2.cpu().numpy(), r2.numpy()[..., 0], rtol=0, atol=ATT_TOL)
    # packed and dense entry points are the same computation
    p_packed, _, _ = eng.embed(centers, labels, 10)
    p_dense, _, _ = eng.embed_dense(dense, 10)
    assert torch.equal(p_packed, p_dense)


def test_dense_general_sem_features(eng, oracle, oracle_sd):
    """The dense entry accepts arbitrary (not one-hot) semantic channels like the reference."""
    rng = np.random.default_rng(5)
    from sg_pr_amd import synth
    centers, labels, _ = synth.make_graphs(8, 64, 20, 50, 9)
    dense = synth.dense_features(centers, labels)
    real = labels >= 0
    soft = rng.dirichlet(np.ones(12) * 0.3, size=labels.shape).astype(np.float32)   # [G,N,12]
    dense[:, 3:, :] = np.where(real[:, None, :], soft.transpose(0, 2, 1), 0.0)
    dense_t = torch.from_numpy(dense)
    p, a, e = eng.embed_dense(dense_t, 10, want_att=True, want_emb=True)
    rp, ra, re = oracle.embed(oracle_sd, dense_t, 10)
    s = eng.score_pairs(p[0::2].contiguous(), p[1::2].contiguous()).cpu()
    rs = oracle.score_from_pooled(oracle_sd, rp[0::2], rp[1::2])
    assert (s - rs).abs().max().item() <= SCORE_TOL


def test_stress_shape_subset_and_invariances(eng, oracle, oracle_sd):
    """Config 5 shape (N=256, k=20): a subset against the oracle, plus size-independent
    properties on a larger batch: batch-position invariance and pair-list == all-pairs."""
    from sg_pr_amd import synth
    centers, labels, _ = synth.config5_pairs(seed=1, batch=64)
    pooled, att, _ = eng.embed(centers, labels, 20, want_att=True)
    dense = torch.from_numpy(synth.dense_features(centers[:16], labels[:16]))
    rp, ra, _ = oracle.embed(oracle_sd, dense, 20)
    rs = oracle.score_from_pooled(oracle_sd, rp[0::2], rp[1::2])
    s = eng.score_pairs(pooled[0:16:2].contiguous(), pooled[1:16:2].contiguous()).cpu()
    print("N=256 max|dscore| =", (s - rs).abs().max().item())
    assert (s - rs).abs().max().item() <= SCORE_TOL
    np.testing.assert_allclose(att[:16].cpu().numpy(), ra.numpy(), rtol=0, atol=ATT_TOL)
    # batch invariance: reversed graph order gives bit-identical per-graph results
    p_rev, _, _ = eng.embed(centers[::-1].copy(), labels[::-1].copy(), 20)
    assert torch.equal(p_rev.flip(0), pooled)
    # all-pairs rectangle == pair list over the same index pairs
    m = eng.score_all_pairs(pooled[:40], pooled)
    ii, jj = torch.meshgrid(torch.arange(40, dtype=torch.int32), torch.arange(128, dtype=torch.int32), indexing="ij")
    lst = eng.score_pairs(pooled, pooled, ii.reshape(-1), jj.reshape(-1)).view(40, 128)
    np.testing.assert_allclose(m.cpu().numpy(), lst.cpu().numpy(), rtol=0, atol=2e-5)  # different summation order


def test_all_pairs_matrix_vs_oracle_and_f1(eng, oracle, oracle_sd):
    """KITTI-like sequence (config 3 generator, reduced M): the dense matrix equals the oracle's
    faithful per-pair evaluation, and F1-max computed from both agrees."""
    from sg_pr_amd import synth, metrics, allpairs
    centers, labels, _, poses = synth.kitti_like_sequence(num_graphs=96, node_num=100, seed=4)
    pooled, _, _ = eng.embed(centers, labels, 10)
    m = eng.score_all_pairs(pooled, pooled).cpu()
    dense = torch.from_numpy(synth.dense_features(centers, labels))
    rp, _, _ = oracle.embed(oracle_sd, dense, 10)
    rm = oracle.score_all_pairs(oracle_sd, rp, rp)
    err = (m - rm).abs().max().item()
    print("all-pairs max|dscore| =", err)
    assert err <= SCORE_TOL
    gt, valid = allpairs.ground_truth_mask(allpairs.pose_distance_matrix(poses), 3)
    assert gt[valid].sum() > 0
    f_hip = metrics.f1_max(gt[valid].numpy(), m[valid].numpy())
    f_ref = oracle.f1_max(gt[valid].numpy(), rm[valid].numpy())
    print("F1-max hip/oracle:", f_hip, f_ref)
    assert abs(f_hip - f_ref) <= 1e-6          # |dF1| <= 1e-6 given score parity (observed: 0)


def test_reference_api_drop_in(golden_dir, ckpt_path):
    """sg_net.SG / SGTrainer entry points reproduce the reference's own outputs."""
    from sg_pr_amd import sg_net
    from sg_pr_amd.parser_sg import sgpr_args
    g = np.load(os.path.join(golden_dir, "kitti3_n100_k10.npz"))
    args = sgpr_args()
    args.model = ckpt_path
    trainer = sg_net.SGTrainer(args, False)
    names = [str(n) for n in g["names"]]
    batch = [[os.path.join(golden_dir, "data", names[i] + ".json"),
              os.path.join(golden_dir, "data", names[j] + ".json")] for i, j in g["pair_ij"]]
    pred, gt = trainer.eval_batch_pair(batch)
    assert pred.dtype == np.float32 and gt.dtype == np.float64
    np.testing.assert_allclose(pred, g["eval_batch_pred"], rtol=0, atol=SCORE_TOL)
    np.testing.assert_array_equal(gt, g["eval_batch_gt"])
    # SG.forward on the dense dictionary (eval_pair.py path)
    feats = torch.from_numpy(g["features"])
    data = {"features_1": torch.stack([feats[i] for i, _ in g["pair_ij"]]),
            "features_2": torch.stack([feats[j] for _, j in g["pair_ij"]])}
    score, a1, a2 = trainer.model(data)
    assert score.shape == (9,) and a1.shape == (9, 100, 1) and a2.shape == (9, 100, 1)
    np.testing.assert_allclose(score.cpu().numpy(), g["scores"], rtol=0, atol=SCORE_TOL)
    e = trainer.model.dgcnn_conv_pass(feats)
    assert e.shape == (3, 100, 32)
    np.testing.assert_allclose(e.cpu().numpy(), g["emb"], rtol=0, atol=EMB_TOL)
    from sg_pr_amd.utils import process_pair
    p, w1, w2 = trainer.eval_pair(process_pair(batch[2]))
    assert abs(p[0] - g["scores"][2]) <= SCORE_TOL and w1.shape == (100,)
    # training mode is refused loudly (BN folded)
    trainer.model.train()
    with pytest.raises(RuntimeError):
        trainer.model(data)
    trainer.model.eval()


def test_error_codes(eng):
    from sg_pr_amd.engine import SgprError
    c = torch.zeros(1, 1025, 3)                      # > SGPR_ANY_MAX_NODES (257..1024 run on the any-shape kernel)
    l = torch.zeros(1, 1025, dtype=torch.int32)
    with pytest.raises(SgprError) as ei:
        eng.embed(c, l, 10)
    assert ei.value.code == -3
    with pytest.raises(SgprError) as ei:
        eng.embed(c[:, :100], l[:, :100], 65)   # > SGPR_ANY_MAX_K
    assert ei.value.code == -4
    with pytest.raises(SgprError) as ei:
        eng.embed(c[:, :8], l[:, :8], 10)       # K > node_num
    assert ei.value.code == -4
    bad = torch.full((2, 32), 3, dtype=torch.int32)
    bad[1, 5] = 12                                   # label outside 0..11 (reference: KeyError)
    eng.embed(torch.zeros(2, 32, 3), bad, 10)
    with pytest.raises(SgprError) as ei:
        eng.check_status()
    assert ei.value.code == -5
    eng.check_status()                               # flag cleared
    # empty batch is a no-op
    p, _, _ = eng.embed(torch.zeros(0, 64, 3), torch.zeros(0, 64, dtype=torch.int32), 10)
    assert p.shape == (0, 32)
    # pooled rows of another width would be read past their end: refused by the binding
    with pytest.raises(ValueError):
        eng.score_pairs(torch.zeros(4, 16), torch.zeros(4, 16))
    with pytest.raises(ValueError):
        eng.score_all_pairs(torch.zeros(4, 32), torch.zeros(4, 31))


def test_odd_sizes(eng, oracle, oracle_sd):
    """Ragged shapes: node_num not a multiple of 16, K not in the tuned set."""
    from sg_pr_amd import synth
    for n, k in ((50, 7), (23, 5), (130, 16), (200, 32)):
        centers, labels, _ = synth.make_graphs(6, n, max(1, n // 3), n - k, 100 + n)
        p, a, _ = eng.embed(centers, labels, k, want_att=True)
        dense = torch.from_numpy(synth.dense_features(centers, labels))
        rp, ra, _ = oracle.embed(oracle_sd, dense, k)
        s = eng.score_pairs(p[0::2].contiguous(), p[1::2].contiguous()).cpu()
        rs = oracle.score_from_pooled(oracle_sd, rp[0::2], rp[1::2])
        assert (s - rs).abs().max().item() <= SCORE_TOL, (n, k)
        np.testing.assert_allclose(a.cpu().numpy(), ra.numpy(), rtol=0, atol=ATT_TOL)


def test_shard_invariance_bitwise(eng):
    """Every score depends on its two graphs only: any split of the work gives bit-identical results
    (the property the multi-GPU row sharding relies on)."""
    from sg_pr_amd import synth
    centers, labels, _, _ = synth.kitti_like_sequence(num_graphs=203, node_num=100, seed=9)
    full, _, _ = eng.embed(centers, labels, 10)
    parts = torch.cat([eng.embed(centers[a:b], labels[a:b], 10)[0] for a, b in ((0, 77), (77, 78), (78, 203))])
    assert torch.equal(full, parts)
    m = eng.score_all_pairs(full, full)
    blocks = torch.cat([eng.score_all_pairs(full[a:b].contiguous(), full) for a, b in ((0, 26), (26, 102), (102, 203))])
    assert torch.equal(m, blocks)


def test_cli_counterparts(tmp_path, golden_dir, ckpt_path, capsys):
    """eval_pair / eval_batch counterparts on the shipped graphs (BASELINE config 1 on the GPU engine)."""
    from sg_pr_amd import eval_pair, eval_batch
    data = os.path.join(golden_dir, "data")
    (tmp_path / "00.txt").write_text("0.json 250.json\n0.json 3.json\n250.json 250.json\n3.json 0.json\n")
    cfg = tmp_path / "config.yml"
    cfg.write_text("""
common: {model: "%s", cuda: "0", batch_size: 128, p_thresh: 3, graph_pairs_dir: "%s", pair_list_dir: '%s'}
arch: {keep_node: 1, filters_1: 64, filters_2: 64, filters_3: 32, tensor_neurons: 16, bottle_neck_neurons: 16, K: 10}
train: {epochs: 500, train_sequences: ['00'], eval_sequences: ["08"], dropout: 0, learning_rate: 0.001,
        weight_decay: 0.0005, gpu: 0, logdir: "./logs_k10", node_num: 100}
eva_batch: {sequences: ["00"], output_path: "%s", show: False}
eva_pair: {pair_file: ["%s/0.json", "%s/250.json"]}
""" % (ckpt_path, data, tmp_path, tmp_path / "eva", data, data))
    score = eval_pair.main([str(cfg)])
    out = capsys.readouterr().out
    assert "Score:" in out and abs(float(score) - 1.3489922e-06) < 1e-6
    res = eval_batch.main([str(cfg)])
    g = np.load(os.path.join(golden_dir, "kitti3_n100_k10.npz"))
    pred = np.load(tmp_path / "eva" / "00_DL_db.npy")
    np.testing.assert_allclose(pred, [g["scores"][2], g["scores"][1], g["scores"][8], g["scores"][3]], atol=SCORE_TOL)
    np.testing.assert_array_equal(np.load(tmp_path / "eva" / "00_gt_db.npy"), [0, 1, 1, 1])
    assert res["00"] == 1.0


def test_node_cap_is_bitwise_neutral_and_loud(eng):
    """sgpr_embed_capped: same bits as the uncapped launch; a broken promise yields NaN + SGPR_E_NODES."""
    from sg_pr_amd import synth
    from sg_pr_amd.engine import SgprError
    # 150 graphs: at most one per CU, the plan ignores the cap (the promise is still enforced);
    # 600 graphs: the capped plan runs (256-thread workgroups, three per CU, key matrix sharing the A region)
    for num_graphs in (150, 600):
        centers, labels, _, _ = synth.kitti_like_sequence(num_graphs=num_graphs, node_num=100, seed=21)
        cap = eng.node_cap_of(centers, labels, 10)
        assert cap == int(synth.effective_nodes(centers, labels, 10).max()) <= 61
        full, att0, _ = eng.embed(centers, labels, 10, want_att=True)
        capped, att1, _ = eng.embed(centers, labels, 10, want_att=True, node_cap=cap)
        assert torch.equal(full, capped) and torch.equal(att0, att1)
        eng.check_status()
        bad, _, _ = eng.embed(centers, labels, 10, node_cap=cap - 1)
        assert torch.isnan(bad).any() and not torch.isnan(bad).all()
        with pytest.raises(SgprError) as ei:
            eng.check_status()
        assert ei.value.code == -3


def test_ordered_embed_is_bitwise_equal_and_loud(eng):
    """sgpr_embed_ordered: any launch order gives the same bits as the plain launch for every output; a subset
    leaves the other rows untouched; a broken node_cap promise is NaN + SGPR_E_NODES."""
    from sg_pr_amd import synth
    from sg_pr_amd.engine import SgprError
    centers, labels, _ = synth.make_graphs(700, 100, 3, 47, seed=5)          # node_cap <= 48: 192-thread workgroups
    full, att0, emb0 = eng.embed(centers, labels, 10, want_att=True, want_emb=True)
    order, cap = eng.size_order(centers, labels, 10)
    assert cap == eng.node_cap_of(centers, labels, 10) <= 48 and sorted(order.tolist()) == list(range(700))
    b, att1, emb1 = eng.embed(centers, labels, 10, want_att=True, want_emb=True, node_cap=cap, order=order)
    eng.check_status()
    assert torch.equal(full, b) and torch.equal(att0, att1) and torch.equal(emb0, emb1)
    # K = 20 > 16 (the other kernel instance), node_num = 256, generic plans
    c2, l2, _ = synth.make_graphs(300, 256, 5, 230, seed=6)
    f2 = eng.embed(c2, l2, 20)[0]
    o2, cap2 = eng.size_order(c2, l2, 20)
    assert torch.equal(f2, eng.embed(c2, l2, 20, node_cap=cap2, order=o2)[0])
    eng.check_status()
    # subset: only the listed graphs are written
    sub = order[::7].contiguous()
    part = eng.embed(centers, labels, 10, node_cap=cap, order=sub)[0]
    assert torch.equal(part[sub.long()], full[sub.long()])
    # broken promise
    bad = eng.embed(centers, labels, 10, node_cap=cap - 1, order=order)[0]
    assert torch.isnan(bad).any() and not torch.isnan(bad).all()
    with pytest.raises(SgprError) as ei:
        eng.check_status()
    assert ei.value.code == -3


def test_plain_embed_hands_oversize_graphs_on(eng):
    """sgpr_embed without a node_cap promise (the reference has no such prerequisite, sg_net.py:503-525) runs the 64-row
    layout when there are more graphs than CUs; a graph with more processed slots is embedded in the same call by the
    instance sized for node_num - inside the second pass (node_num <= 128) or in a launch of its own (node_num 256) -
    with the bits of the capped launch, and is no error.  An explicit promise that is broken stays loud."""
    from sg_pr_amd import synth
    from sg_pr_amd.engine import SgprError
    for num, n, lo, hi, kitti in ((1500, 100, 25, 70, True), (700, 100, 40, 85, False), (900, 128, 30, 110, False),
                                  (600, 256, 20, 120, False), (400, 100, 3, 60, False)):
        centers, labels, _ = synth.make_graphs(num, n, lo, hi, 900 + n + hi, kitti_like=kitti)
        eff = eng.processed_slots(centers, labels, 10)
        cap = int(eff.max())
        plain, att0, emb0 = eng.embed(centers, labels, 10, want_att=True, want_emb=True)
        eng.check_status()                                            # nothing to report
        capped, att1, emb1 = eng.embed(centers, labels, 10, want_att=True, want_emb=True, node_cap=cap)
        eng.check_status()
        assert torch.equal(plain, capped) and torch.equal(att0, att1) and torch.equal(emb0, emb1), (num, n, cap)
        assert not torch.isnan(plain).any()
        if hi > 64:
            assert int((eff > 64).sum()) > 0                          # the hand-over ran
    # ragged store and explicit launch order, still without a promise
    centers, labels, _ = synth.make_graphs(1200, 100, 30, 80, 77)
    rc, rl, ro = eng.to_ragged(centers, labels)
    order, cap = eng.size_order(centers, labels, 10)
    ref = eng.embed(centers, labels, 10, node_cap=cap)[0]
    assert torch.equal(eng.embed_ragged(rc, rl, ro, 100, 10)[0], ref)
    assert torch.equal(eng.embed(centers, labels, 10, order=order)[0], ref)
    eng.check_status()
    # resident tensors: the binding launches in the device-made order it remembers per tensor pair (same bits), and an
    # in-place change of the data makes it compute the order again
    dc, dl = torch.from_numpy(centers).cuda(), torch.from_numpy(labels).cuda()
    assert torch.equal(eng.embed(dc, dl, 10)[0], ref) and torch.equal(eng.embed(dc, dl, 10, auto_order=False)[0], ref)
    n_cached = len(eng._order_cache)
    assert torch.equal(eng.embed(dc, dl, 10)[0], ref) and len(eng._order_cache) == n_cached      # hit
    dl[5, 70:] = -1
    dc[5, 70:] = 0.0
    changed = eng.embed(dc, dl, 10)[0]
    assert eng._order_cache[-1][2][3:5] == (dc._version, dl._version)                            # recomputed
    torch.cuda.synchronize()
    assert eng._cached_order(dc, dl, 10)[1] == int(eng.processed_slots(dc, dl, 10).max())       # ... and its node_cap arrives
    assert torch.equal(changed, eng.embed(dc.cpu().numpy(), dl.cpu().numpy(), 10)[0])
    eng.check_status()
    # an entry belongs to tensor OBJECTS: another data set of the same shape (whatever address the allocator gives it) is a
    # miss, so a remembered node_cap can never be promised for data it was not computed from
    other_c, other_l, _ = synth.make_graphs(1200, 100, 60, 95, 78)
    del dc, dl
    oc, ol = torch.from_numpy(other_c).cuda(), torch.from_numpy(other_l).cuda()
    assert torch.equal(eng.embed(oc, ol, 10)[0], eng.embed(other_c, other_l, 10)[0])
    torch.cuda.synchronize()
    assert torch.equal(eng.embed(oc, ol, 10)[0], eng.embed(other_c, other_l, 10)[0])               # (now with its own node_cap)
    eng.check_status()
    # a promise above 64 slots: two tiers as well (the small graphs on the 64-row layout, the others on the instance sized
    # for the promise) - same bits; broken: loud
    true_cap = int(eng.processed_slots(centers, labels, 10).max())
    assert true_cap > 70
    assert torch.equal(eng.embed(centers, labels, 10, node_cap=true_cap)[0], ref)
    eng.check_status()
    bad = eng.embed(centers, labels, 10, node_cap=70)[0]
    assert torch.isnan(bad).any() and not torch.isnan(bad).all()
    with pytest.raises(SgprError) as ei:
        eng.check_status()
    assert ei.value.code == -3
    # a promise is a promise
    bad = eng.embed(centers, labels, 10, node_cap=64)[0]
    assert torch.isnan(bad).any()
    with pytest.raises(SgprError) as ei:
        eng.check_status()
    assert ei.value.code == -3


def test_size_order_on_the_device(eng):
    """sgpr_size_order (processed slots -> node_cap + stable largest-first order, two launches on the device) against
    the torch form of the same rule, padded arrays and ragged stores; labels outside the checkpoint's classes count as
    padding there as they do in the embed kernel (which reports them)."""
    from sg_pr_amd import synth
    for num, n, lo, hi, k in ((4541, 100, 25, 60, 10), (700, 64, 17, 64, 10), (1000, 256, 100, 236, 20), (3, 100, 1, 5, 10),
                              (70000, 24, 1, 24, 4), (513, 100, 0, 100, 10)):
        centers, labels, _ = synth.make_graphs(num, n, max(lo, 1), hi, 5 + num)
        if lo == 0:
            labels[::7] = -1                                          # empty graphs
            centers[::7] = 0.0
        want_order, want_cap = eng.size_order_torch(centers, labels, k)
        order, cap = eng.size_order(centers, labels, k)
        assert cap == want_cap and torch.equal(order, want_order), (num, n)
        info = eng.size_order_device(torch.from_numpy(centers).cuda(), torch.from_numpy(labels).cuda(), None, n, k)[1]
        assert info.tolist() == [want_cap, int((eng.processed_slots(centers, labels, k) > 64).sum())]
        rc, rl, ro = eng.to_ragged(centers, labels)
        r_order, r_cap = eng.ragged_order(ro, n, k)
        off = torch.as_tensor(ro)
        cnt = off[1:] - off[:-1]
        m = n - cnt
        eff = cnt + torch.where((m >= k) & (m > 1), torch.ones_like(m), m)
        assert r_cap == int(eff.max()) and torch.equal(r_order.cpu(), torch.argsort(eff, descending=True, stable=True).to(torch.int32))


def test_device_f1_max_and_counts(eng):
    """the score matrix stays on the GPU.  sgpr_pair_positives / sgpr_pair_threshold_counts == their numpy
    restatements (poses and explicit labels, sharded rows, ranking of the negatives), and F1-max / ROC area equal the
    sorted host computation - with the full threshold budget and with one that forces several refinement passes."""
    from sg_pr_amd import synth, metrics, allpairs
    centers, labels, _, poses = synth.kitti_like_sequence(num_graphs=700, node_num=100, seed=9)
    order, cap = eng.size_order(centers, labels, 10)
    pooled = eng.embed(centers, labels, 10, node_cap=cap, order=order)[0]
    m = eng.score_all_pairs(pooled, pooled)
    xz = allpairs.pose_xz(poses)
    d = torch.cdist(xz.double(), xz.double())
    gt = torch.where(d <= 3, 1, torch.where(d >= 20, 0, -1)).to(torch.int8)
    mh = m.cpu().numpy()
    pos_host, count_host = metrics.counts_of(mh, gt.numpy())
    pos, bad = eng.pair_positives(m, pose_xz=xz)
    assert bad == 0 and pos.numel() == int((gt == 1).sum())
    np.testing.assert_array_equal(np.sort(pos.cpu().numpy()), np.sort(pos_host))
    u, mult = np.unique(pos_host, return_counts=True)
    above = np.concatenate((np.cumsum(mult[::-1])[::-1], [0])).astype(np.int64)
    for step in (1, 3, 50):                              # thresholds = every step-th distinct positive score
        thr = u[::step]
        if thr.size > metrics.MAX_THRESHOLDS:
            continue
        c_dev, bad, r_dev = eng.pair_threshold_counts(m, thr, pose_xz=xz, rank=(u, step, above))
        c_host, r_host = count_host(thr, (u, step, above))
        assert bad == 0 and r_dev == r_host
        np.testing.assert_array_equal(c_dev, c_host)
        assert int(c_dev.sum()) == int((gt == 0).sum())
    # explicit labels, a row shard (row0 != 0, odd sizes), thresholds that are not positive scores, none at all
    thr = np.linspace(0.0, 1.0, 777, dtype=np.float32)
    c1, _, _ = eng.pair_threshold_counts(m, thr, pose_xz=xz)
    c2, _, _ = eng.pair_threshold_counts(m, thr, gt=gt)
    np.testing.assert_array_equal(c1, c2)
    np.testing.assert_array_equal(c1, count_host(thr, None)[0])
    c3, _, _ = eng.pair_threshold_counts(m[101:358], thr, row0=101, pose_xz=xz)
    np.testing.assert_array_equal(c3, metrics.counts_of(mh[101:358], gt[101:358].numpy())[1](thr, None)[0])
    p3, _ = eng.pair_positives(m[101:358], row0=101, pose_xz=xz)
    np.testing.assert_array_equal(np.sort(p3.cpu().numpy()), np.sort(mh[101:358][gt[101:358].numpy() == 1]))
    c4, _, _ = eng.pair_threshold_counts(m, np.zeros(0, dtype=np.float32), pose_xz=xz)
    assert c4.tolist() == [int((gt == 0).sum())]
    valid = gt >= 0
    f_host = metrics.f1_max(gt[valid].numpy(), m.cpu()[valid].numpy())
    a_host = metrics.roc_auc(gt[valid].numpy(), m.cpu()[valid].numpy())
    f_dev, a_dev, passes = metrics.pr_roc_device(eng, m, pose_xz=xz)
    print("F1-max device / host:", f_dev, f_host, "AUC", a_dev, a_host, "passes", passes, "positives", pos.numel())
    assert abs(f_dev - f_host) < 1e-12 and abs(a_dev - a_host) < 1e-12 and passes <= 2
    distinct, count_fn = metrics._device_fns(eng, m, xz, 3.0, 20.0, None, 0)
    assert np.array_equal(distinct[0], u) and np.array_equal(distinct[1], mult)
    f_small, a_small, p_small = metrics.pr_roc_from_counts(None, count_fn, max_thresholds=64, distinct=distinct)
    assert abs(f_small - f_host) < 1e-12 and abs(a_small - a_host) < 1e-12 and p_small >= 2
    assert metrics.f1_max_device(eng, m, pose_xz=xz)[0] == f_dev
    assert abs(metrics.roc_auc_device(eng, m, pose_xz=xz) - a_host) < 1e-12
    # ragged shapes: M not a multiple of 4, rows with a padded leading dimension, one row, random labels incl. ignored
    rng = np.random.default_rng(3)
    for rows, cols, ld in ((37, 333, 340), (1, 5, 5), (129, 1023, 1023), (3, 2, 8)):
        buf = torch.rand(rows, ld, generator=torch.Generator().manual_seed(rows)).cuda()
        sc = buf[:, :cols]
        lab = torch.from_numpy(rng.integers(-1, 2, size=(rows, cols)).astype(np.int8))
        p_host, c_host = metrics.counts_of(sc.cpu().numpy(), lab.numpy())
        p_dev, bad = eng.pair_positives(sc, gt=lab)
        assert bad == 0
        np.testing.assert_array_equal(np.sort(p_dev.cpu().numpy()), np.sort(p_host))
        for t in (0, 1, 17, 1000):
            thr = np.sort(rng.random(t).astype(np.float32))
            c_dev, _, _ = eng.pair_threshold_counts(sc, thr, gt=lab)
            np.testing.assert_array_equal(c_dev, c_host(thr, None)[0])
        if p_host.size and (lab == 0).any():
            f, a_, _ = metrics.pr_roc_device(eng, sc, gt=lab)
            keep = lab.numpy().ravel() >= 0
            assert abs(f - metrics.f1_max(lab.numpy().ravel()[keep], sc.cpu().numpy().ravel()[keep])) < 1e-12
            assert abs(a_ - metrics.roc_auc(lab.numpy().ravel()[keep], sc.cpu().numpy().ravel()[keep])) < 1e-12
    # negative scores are refused, whichever class they belong to
    bad_m = m.clone()
    bad_m[3, 5] = -0.25
    one_positive = torch.zeros_like(gt)
    one_positive[0, 0] = 1
    for labels_of in (torch.ones_like(gt), one_positive):
        with pytest.raises(ValueError):
            metrics.f1_max_device(eng, bad_m, gt=labels_of)


def test_topk_rows_loop_closures(eng):
    """best matches per query row outside a temporal window == torch.topk on the masked matrix."""
    g = torch.Generator().manual_seed(3)
    m = torch.rand(301, 517, generator=g)
    m[:, ::7] = m[:, 3:4]                                   # ties: lowest column must win
    md = m.cuda()
    for k in (1, 4, 8, 16):
        for window, row0 in ((-1, 0), (10, 0), (50, 120)):
            vals, idx = eng.topk_rows(md, k=k, row0=row0, window=window)
            ref = m.clone()
            if window >= 0:
                rr = torch.arange(301).view(-1, 1) + row0
                cc = torch.arange(517).view(1, -1)
                ref[(rr - cc).abs() <= window] = -float("inf")
            # stable descending sort == (value desc, column asc)
            order = torch.sort(ref, dim=1, descending=True, stable=True)
            np.testing.assert_array_equal(idx.cpu().numpy(), order.indices[:, :k].numpy().astype(np.int32))
            np.testing.assert_array_equal(vals.cpu().numpy(), order.values[:, :k].numpy())
    # fewer than k qualifying columns -> -1
    vals, idx = eng.topk_rows(md[:5, :6].contiguous(), k=8, window=1)
    assert (idx.cpu()[:, -1] == -1).all() and torch.isinf(vals.cpu()[:, -1]).all()


def test_custom_ops_match_engine(eng, ckpt_path):
    """torch.ops.sgpr.* == the Engine methods they wrap (same C-ABI calls), bit for bit."""
    from sg_pr_amd import ops, synth, engine  # noqa: F401
    sd = torch.load(ckpt_path, map_location="cpu")
    blob = torch.from_numpy(engine.blob_from_state_dict(sd)).cuda()
    centers, labels, _ = synth.config2_pairs(seed=1, batch=8)
    c, l = torch.from_numpy(centers).cuda(), torch.from_numpy(labels).cuda()
    pooled, att = torch.ops.sgpr.embed(c, l, blob, 10)
    p0, a0, _ = eng.embed(c, l, 10, want_att=True)
    assert torch.equal(pooled, p0) and torch.equal(att, a0)
    assert torch.equal(torch.ops.sgpr.score_all_pairs(pooled, pooled, blob), eng.score_all_pairs(p0, p0))
    assert torch.equal(torch.ops.sgpr.score_pairs(pooled[0::2].contiguous(), pooled[1::2].contiguous(), blob),
                       eng.score_pairs(p0[0::2].contiguous(), p0[1::2].contiguous()))
    dense = torch.from_numpy(synth.dense_features(centers, labels)).cuda()
    s1, x1, y1 = torch.ops.sgpr.forward_dense(dense[0::2].contiguous(), dense[1::2].contiguous(), blob, 10)
    s0, x0, y0 = eng.forward_dense(dense[0::2].contiguous(), dense[1::2].contiguous(), 10)
    assert torch.equal(s1, s0) and torch.equal(x1, x0) and torch.equal(y1, y0)


def test_lean_plans_of_every_shape(eng, oracle, oracle_sd):
    """Capped plans with 128 / 192 / 256 threads, K <= 16 and K > 16 instances: bit-identical to the uncapped
    launch (which test_odd_sizes & co. hold against the oracle), plus one direct oracle check per shape."""
    from sg_pr_amd import synth
    for n, k, lo, hi in ((100, 10, 3, 14), (40, 10, 5, 28), (100, 10, 33, 47), (64, 20, 10, 40), (64, 5, 40, 59),
                         (100, 32, 20, 60), (24, 3, 2, 20)):
        centers, labels, _ = synth.make_graphs(300, n, lo, hi, 1000 + n + k)
        order, cap = eng.size_order(centers, labels, k)
        assert cap <= 64
        full, att0, _ = eng.embed(centers, labels, k, want_att=True)
        lean, att1, _ = eng.embed(centers, labels, k, want_att=True, node_cap=cap, order=order)
        eng.check_status()
        assert torch.equal(full, lean) and torch.equal(att0, att1), (n, k, lo, hi)
        sub = slice(0, 12)
        rp, _, _ = oracle.embed(oracle_sd, torch.from_numpy(synth.dense_features(centers[sub], labels[sub])), k)
        s = eng.score_pairs(lean[sub][0::2].contiguous(), lean[sub][1::2].contiguous()).cpu()
        rs = oracle.score_from_pooled(oracle_sd, rp[0::2], rp[1::2])
        assert (s - rs).abs().max().item() <= SCORE_TOL, (n, k)


def test_sequence_evaluation_end_to_end(tmp_path, golden_dir, ckpt_path):
    """Packed store -> embed once -> dense matrix -> device F1-max + loop closures, through the module CLI."""
    from sg_pr_amd import graph_store, metrics
    data = os.path.join(golden_dir, "data")
    os.makedirs(tmp_path / "graphs" / "00")
    for f in os.listdir(data):
        with open(os.path.join(data, f)) as src, open(tmp_path / "graphs" / "00" / f, "w") as dst:
            dst.write(src.read())
    cfg = tmp_path / "config.yml"
    cfg.write_text("""
common: {model: "%s", cuda: "0", batch_size: 128, p_thresh: 3, graph_pairs_dir: "%s", pair_list_dir: '%s'}
arch: {keep_node: 1, filters_1: 64, filters_2: 64, filters_3: 32, tensor_neurons: 16, bottle_neck_neurons: 16, K: 10}
train: {epochs: 500, train_sequences: ['00'], eval_sequences: ["08"], dropout: 0, learning_rate: 0.001,
        weight_decay: 0.0005, gpu: 0, logdir: "./logs_k10", node_num: 100}
eva_batch: {sequences: ["00"], output_path: "%s", show: False}
eva_pair: {pair_file: ["%s/0.json", "%s/250.json"]}
""" % (ckpt_path, tmp_path / "graphs", tmp_path, tmp_path / "eva", data, data))
    res = graph_store.main([str(cfg)])
    g = np.load(os.path.join(golden_dir, "kitti3_n100_k10.npz"))
    # frames in natural order 0, 3, 250 == the golden's graph order; 9 ordered pairs row-major
    seq = graph_store.PackedSequence.load(str(tmp_path / "eva" / "00_packed.npz"))
    assert seq.names == ["0.json", "3.json", "250.json"]
    lc = np.load(tmp_path / "eva" / "00_loop_closures.npy")
    assert lc.shape == (3, 3)
    # reference values: golden scores + float64 pose distances
    scores = g["scores"].reshape(3, 3)
    d = np.sqrt((seq.poses[:, None, 3] - seq.poses[None, :, 3]) ** 2 + (seq.poses[:, None, 11] - seq.poses[None, :, 11]) ** 2)
    valid = (d <= 3) | (d >= 20)
    want = metrics.f1_max((d <= 3)[valid].astype(np.float64), scores[valid])
    assert abs(res["00"] - want) < 1e-9


def test_super_node_branch_equals_generic_on_every_entry_point(eng):
    """Packed and dense (one-hot) inputs take the super-node semantic branch; ablation bit 12 forces the generic branch
    on the same kernel: all three give the same bits, for embed, embed_dense and the drop-in forward_dense."""
    from sg_pr_amd import synth
    centers, labels, _ = synth.config2_pairs(seed=11, batch=150)              # 300 graphs, N = 64
    dense = torch.from_numpy(synth.dense_features(centers, labels)).cuda()
    order, cap = eng.size_order(centers, labels, 10)
    fast_p = eng.embed(centers, labels, 10, want_att=True, want_emb=True, node_cap=cap, order=order)
    fast_d = eng.embed_dense(dense, 10, want_att=True, want_emb=True)
    fwd = eng.forward_dense(dense[0::2].contiguous(), dense[1::2].contiguous(), 10)
    eng.set_skip_mask(256 + 4096)
    try:
        gen_p = eng.embed(centers, labels, 10, want_att=True, want_emb=True, node_cap=cap, order=order)
        gen_d = eng.embed_dense(dense, 10, want_att=True, want_emb=True)
        fwd_g = eng.forward_dense(dense[0::2].contiguous(), dense[1::2].contiguous(), 10)
    finally:
        eng.set_skip_mask(0)
    for a, b, c, d in zip(fast_p, fast_d, gen_p, gen_d):
        assert torch.equal(a, b) and torch.equal(a, c) and torch.equal(a, d)
    for a, b in zip(fwd, fwd_g):
        assert torch.equal(a, b)
    # general (not one-hot) semantic rows keep working: generic branch, different results from the one-hot ones
    noisy = dense.clone()
    noisy[:, 3:, :] += 0.25
    assert not torch.equal(eng.embed_dense(noisy, 10)[0], fast_d[0])


def test_label_lookup_layer_is_bitwise_generic(eng):
    """The label-lookup first semantic layer (packed input) gives the bits of the generic kNN/EdgeConv layer, which
    the debug-dump instance still runs - on KITTI-like graphs, on graphs with fewer label-mates than K, with every
    label present, and in the lean and the latency plans."""
    from sg_pr_amd import synth
    for num, n, lo, hi, kitti in ((40, 100, 25, 60, True), (300, 100, 25, 60, True), (300, 64, 17, 50, False),
                                  (300, 100, 40, 85, False), (24, 160, 60, 140, False), (24, 256, 100, 236, False)):
        centers, labels, _ = synth.make_graphs(num, n, lo, hi, 77 + num + n, kitti_like=kitti)
        k = 20 if n == 256 else 10
        order, cap = eng.size_order(centers, labels, k)
        prod, att, emb = eng.embed(centers, labels, k, want_att=True, want_emb=True, node_cap=cap, order=order)
        dbg = eng.embed(centers, labels, k, debug=True)
        assert torch.equal(prod, dbg[0]) and torch.equal(att, dbg[1]) and torch.equal(emb, dbg[2]), (num, n)


def test_full_size_kitti00_properties(eng, oracle, oracle_sd):
    """BASELINE config 3 at full size (M = 4541, node_num 100, K 10): properties that do not need the oracle on 20 M
    pairs - a random sample of pairs against the oracle's faithful per-pair forward, pair-list == matrix entries,
    row shards bit-identical, device F1-max == sorted host F1-max on a row block, closures consistent."""
    from sg_pr_amd import synth, metrics, allpairs
    centers, labels, _, poses = synth.kitti_like_sequence(4541, 100, 0)
    order, cap = eng.size_order(centers, labels, 10)
    pooled = eng.embed(centers, labels, 10, node_cap=cap, order=order)[0]
    eng.check_status()
    m = eng.score_all_pairs(pooled, pooled)
    assert m.shape == (4541, 4541) and bool(torch.isfinite(m).all()) and float(m.min()) >= 0 and float(m.max()) <= 1
    rng = np.random.default_rng(5)
    # (1) oracle on 48 random ordered pairs
    ii, jj = rng.integers(0, 4541, 48), rng.integers(0, 4541, 48)
    f1 = torch.from_numpy(synth.dense_features(centers[ii], labels[ii]))
    f2 = torch.from_numpy(synth.dense_features(centers[jj], labels[jj]))
    ref = oracle.forward(oracle_sd, f1, f2, 10)[0]
    got = m[torch.from_numpy(ii).cuda(), torch.from_numpy(jj).cuda()].cpu()
    assert (got - ref).abs().max().item() <= SCORE_TOL
    # (2) pair-list kernel == matrix entries
    pi = torch.from_numpy(rng.integers(0, 4541, 8192).astype(np.int32))
    pj = torch.from_numpy(rng.integers(0, 4541, 8192).astype(np.int32))
    lst = eng.score_pairs(pooled, pooled, pi, pj).cpu()
    np.testing.assert_allclose(lst.numpy(), m[pi.long().cuda(), pj.long().cuda()].cpu().numpy(), rtol=0, atol=2e-5)
    # (3) any row shard reproduces its rows bit for bit; embedding a shard alone reproduces its pooled rows
    for lo, hi in ((0, 568), (1703, 2271), (4000, 4541)):
        assert torch.equal(eng.score_all_pairs(pooled[lo:hi].contiguous(), pooled), m[lo:hi])
        o2, c2 = eng.size_order(centers[lo:hi], labels[lo:hi], 10)
        assert torch.equal(eng.embed(centers[lo:hi], labels[lo:hi], 10, node_cap=c2, order=o2)[0], pooled[lo:hi])
    # (4) F1-max: device counts over the whole matrix == sum-consistent with a sorted host computation on a block
    xz = allpairs.pose_xz(poses)
    blk = m[1000:1600]
    f_dev, _ = metrics.f1_max_device(eng, blk, pose_xz=xz, row0=1000)
    d = torch.cdist(xz[1000:1600], xz)
    valid = (d <= 3) | (d >= 20)
    f_host = metrics.f1_max((d <= 3)[valid].numpy(), blk.cpu()[valid].numpy())
    assert abs(f_dev - f_host) < 1e-12
    f_all, passes = metrics.f1_max_device(eng, m, pose_xz=xz)
    assert 0.0 <= f_all <= 1.0 and passes <= 4
    # (5) loop closures: top-1 outside a 50-frame window really is the row maximum there
    vals, idx = eng.topk_rows(m, k=1, window=50)
    r = 2345
    row = m[r].clone()
    row[max(0, r - 50):r + 51] = -1
    assert int(idx[r, 0]) == int(torch.argmax(row)) and float(vals[r, 0]) == float(row.max())


def test_c_demo_matches_engine(tmp_path, eng, ckpt_path):
    """examples/sgpr_demo.c (plain C, no Python / torch in the process) produces the Engine's score matrix bit for bit."""
    import subprocess
    from sg_pr_amd import engine, synth
    from test_host_cpu import _build_c_demo
    exe = _build_c_demo(tmp_path)
    sd = torch.load(ckpt_path, map_location="cpu")
    engine.blob_from_state_dict(sd).tofile(str(tmp_path / "w.f32"))
    centers, labels, _, _ = synth.kitti_like_sequence(37, 100, 12)
    with open(tmp_path / "g.bin", "wb") as f:
        np.array([37, 100, 10], dtype=np.int32).tofile(f)
        centers.astype(np.float32).tofile(f)
        labels.astype(np.int32).tofile(f)
    out = subprocess.run([exe, str(tmp_path / "w.f32"), str(tmp_path / "g.bin"), str(tmp_path / "s.f32")],
                         check=True, capture_output=True, text=True).stdout
    assert "scored 37 x 37" in out
    assert "pair list: 185 pairs" in out and ", 0 differ from the dense matrix" in out        # sgpr_pair_plan + sgpr_score_pair_list
    got = np.fromfile(tmp_path / "s.f32", dtype=np.float32).reshape(37, 37)
    pooled = eng.embed(centers, labels, 10)[0]
    np.testing.assert_array_equal(got, eng.score_all_pairs(pooled, pooled).cpu().numpy())


# ------------------------------------------------------------------ N > 1 code path on ONE GPU (gloo ranks share cuda:0)
def _two_rank_worker(rank, world, port, ckpt, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from sg_pr_amd import synth, allpairs, sg_net
    from sg_pr_amd.parser_sg import sgpr_args
    args = sgpr_args()
    args.model = ckpt
    trainer = sg_net.SGTrainer(args, False)
    centers, labels, _, poses = synth.kitti_like_sequence(203, 100, 6)          # 203 = 102 + 101 rows: uneven shards
    dc, dl = torch.from_numpy(centers).cuda(), torch.from_numpy(labels).cuda()
    scorer = allpairs.AllPairsScorer(model=trainer.model)
    full = scorer.run(dc, dl, chunks=4)                                          # pieces overlapped with scoring
    plain = scorer.run(dc, dl, chunks=1)                                         # plain gather
    block = scorer.run(dc, dl, gather=False)
    f1 = scorer.f1_max(block, poses)
    # pair-list mode (eval_batch.py:30-36) sharded over the ranks: graphs first (60 of the 120 per rank, one all-gather of
    # pooled), then 2600 listed pairs per rank -> the grouped kernel
    from sg_pr_amd import eval_batch
    pairs = [l.split() for l in open(os.path.join(out_dir, "pairs.txt")).read().splitlines()]
    pred, gt = eval_batch.score_pair_list(trainer, pairs)
    # graph-sharded: this rank parsed and embedded ITS half of the distinct graphs (a split of the list alone: nearly all)
    n_graphs = len({p for ab in pairs for p in ab})
    lo_g, hi_g = allpairs.shard_bounds(n_graphs, world, rank)
    assert eval_batch.score_pair_list.last_embedded == hi_g - lo_g <= n_graphs // world + 1
    if rank == 0:
        assert torch.equal(full, plain)
        torch.save({"full": full.cpu(), "f1": f1, "pred": pred, "gt": gt}, os.path.join(out_dir, "two.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _write_graph_dir(tmp_path, n_graphs=120, n_pairs=5200, seed=4):
    """graph JSONs (utils.py:21-38 format) 100 m apart + a shuffled pair list with repeated row graphs"""
    import json
    from sg_pr_amd import synth
    centers, labels, n_real, _ = synth.kitti_like_sequence(n_graphs, 100, seed)
    gdir = tmp_path / "graphs"
    gdir.mkdir()
    for g in range(n_graphs):
        n = int(n_real[g])
        pose = [1.0, 0.0, 0.0, 100.0 * g, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0]
        json.dump({"nodes": labels[g, :n].tolist(), "centers": centers[g, :n].astype(float).tolist(), "pose": pose},
                  open(gdir / ("%d.json" % g), "w"))
    rng = np.random.default_rng(seed)
    ij = rng.integers(0, n_graphs, size=(n_pairs, 2))
    (tmp_path / "pairs.txt").write_text("".join("%s %s\n" % (gdir / ("%d.json" % a), gdir / ("%d.json" % b)) for a, b in ij))
    return [[str(gdir / ("%d.json" % a)), str(gdir / ("%d.json" % b))] for a, b in ij]


@pytest.mark.timeout(600)
def test_two_ranks_on_one_gpu_bitwise_equal_one_rank(tmp_path, ckpt_path):
    """The real AllPairsScorer.run (HIP embed / all-pairs kernels, sharded rows, overlapped gather in 4 pieces and the
    plain gather) with two ranks on cuda:0: the gathered matrix is bit-identical to the single-process one and the
    sharded device F1-max equals the single-rank value.  (RCCL needs one GPU per rank; gloo ranks can share one.)"""
    import torch.multiprocessing as mp
    from sg_pr_amd import synth, allpairs, sg_net, eval_batch
    from sg_pr_amd.parser_sg import sgpr_args
    pairs = _write_graph_dir(tmp_path)
    mp.spawn(_two_rank_worker, args=(2, 29641, ckpt_path, str(tmp_path)), nprocs=2, join=True)
    two = torch.load(os.path.join(str(tmp_path), "two.pt"), weights_only=False)
    args = sgpr_args()
    args.model = ckpt_path
    trainer = sg_net.SGTrainer(args, False)
    centers, labels, _, poses = synth.kitti_like_sequence(203, 100, 6)
    scorer = allpairs.AllPairsScorer(model=trainer.model)
    one = scorer.run(torch.from_numpy(centers).cuda(), torch.from_numpy(labels).cuda())
    assert torch.equal(one.cpu(), two["full"])
    assert scorer.f1_max(one, poses) == two["f1"]
    # the sharded pair list == the single-process one, bit for bit (each rank grouped and scored its own 2600 pairs)
    pred, gt = eval_batch.score_pair_list(trainer, pairs)
    assert len(pred) == 5200 and np.array_equal(pred, two["pred"]) and np.array_equal(gt, two["gt"])
    assert set(np.unique(gt)) == {0.0, 1.0}


def test_sequence_set_equals_per_sequence_runs(ckpt_path):
    """allpairs.SequenceSet (one embed launch for the graphs of several sequences, eval_batch.py:26-36's loop) gives the
    matrices of per-sequence runs bit for bit."""
    from sg_pr_amd import synth, allpairs, sg_net
    from sg_pr_amd.parser_sg import sgpr_args
    args = sgpr_args()
    args.model = ckpt_path
    model = sg_net.SGTrainer(args, False).model
    eng = model.engine()
    seqs = []
    for seed, m in ((1, 150), (2, 37), (3, 301)):
        c, l, _, _ = synth.kitti_like_sequence(m, 100, seed)
        seqs.append((torch.from_numpy(c).cuda(), torch.from_numpy(l).cuda()))
    scorer = allpairs.AllPairsScorer(model=model)
    one = [scorer.run(c, l) for c, l in seqs]
    sset = allpairs.SequenceSet(scorer, seqs)
    order, cap = eng.size_order(sset.centers, sset.labels, 10)
    many = sset.run(embed_fn=lambda c, l: eng.embed(c, l, 10, node_cap=cap, order=order)[0])
    assert [tuple(x.shape) for x in many] == [(150, 150), (37, 37), (301, 301)]
    for a_, b_ in zip(one, many):
        assert torch.equal(a_, b_)
    # ... and so does the per-sequence form of the set (batch_tails=False) and the engine entry point on ragged
    # rectangles (rows != columns, an empty one, outputs with a padded leading dimension, more jobs than one call takes)
    for a_, b_ in zip(one, allpairs.SequenceSet(scorer, seqs, batch_tails=False).run()):
        assert torch.equal(a_, b_)
    pooled = [eng.embed(c, l, 10)[0] for c, l in seqs]
    jobs = [(pooled[0][:17], pooled[2]), (pooled[1], pooled[1][:0]), (pooled[2][5:300], pooled[0]),
            (pooled[1][:1], pooled[1]), (pooled[0], pooled[1], torch.empty(150, 64, device="cuda")[:, :37])]
    jobs.append((pooled[0][:40] * 400.0, pooled[2][:90] * 400.0))     # outside the f16 range: its own exact path, alone
    jobs = jobs + jobs                                   # 12 jobs: two calls of the C entry point
    got = eng.score_all_pairs_multi(jobs)
    for (rows, cols, *_), g_ in zip(jobs, got):
        assert g_.shape == (rows.shape[0], cols.shape[0])
        if g_.numel():
            assert torch.equal(g_, eng.score_all_pairs(rows.contiguous(), cols))


# ------------------------------------------------------------------ config 4 on real RCCL (needs >= 2 GPUs; skips cleanly on a 1-GPU box)
def _rccl_worker(rank, world, port, ckpt, golden, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from sg_pr_amd import synth, allpairs, sg_net, eval_batch
    from sg_pr_amd.parser_sg import sgpr_args
    args = sgpr_args()
    args.model = ckpt
    args.gpu = rank
    args.cuda = str(rank)
    trainer = sg_net.SGTrainer(args, False)
    dev = torch.device("cuda", rank)
    seqs, poses = [], []
    for seed, m in ((6, 403), (7, 131), (8, 7)):                       # uneven shards; 7 graphs < 8 ranks: empty shards
        c, l, _, p = synth.kitti_like_sequence(m, 100, seed)
        seqs.append((torch.from_numpy(c).to(dev), torch.from_numpy(l).to(dev)))
        poses.append(p)
    scorer = allpairs.AllPairsScorer(model=trainer.model)
    full = scorer.run(*seqs[0], chunks=4)                               # pieces shipped over xGMI while the next is scored
    plain = scorer.run(*seqs[0], chunks=1)                              # plain gather
    block = scorer.run(*seqs[0], gather=False)
    f1, auc = scorer.pr_roc(block, poses[0])                            # all_gather of positives + all_reduce of counts
    many = allpairs.SequenceSet(scorer, seqs).run(chunks=4)             # config 4's shape: sequences back to back
    sharded = allpairs.SequenceSet(scorer, seqs).run(gather=False)      # batched tails, matrices left sharded
    # pair-list mode (eval_batch.py:30-36) in shards over RCCL
    data = os.path.join(golden, "data")
    names = ["0.json", "3.json", "250.json"]
    pairs = [[os.path.join(data, a), os.path.join(data, b)] for a in names for b in names] * 3
    pred, gt = eval_batch.score_pair_list(trainer, pairs)
    lo, hi = allpairs.shard_bounds(seqs[1][1].shape[0], world, rank)
    torch.save({"rows": sharded[1].cpu(), "lo": lo, "hi": hi}, os.path.join(out_dir, "rows_r%d.pt" % rank))
    if rank == 0:
        assert torch.equal(full, plain)
        torch.save({"full": full.cpu(), "many": [x.cpu() for x in many], "f1": f1, "auc": auc, "pred": pred, "gt": gt},
                   os.path.join(out_dir, "rccl.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _rccl_single_worker(rank, port, out_path):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    dev = torch.device("cuda", 0)
    t = torch.arange(8, dtype=torch.float32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    parts = [torch.empty(5, dtype=torch.int64, device=dev)]
    dist.all_gather(parts, torch.arange(5, dtype=torch.int64, device=dev))
    b = torch.full((3,), 7.0, device=dev)
    dist.broadcast(b, src=0)
    dist.barrier()
    torch.cuda.synchronize()
    torch.save({"reduced": t.cpu(), "gathered": parts[0].cpu(), "bcast": b.cpu(), "backend": dist.get_backend()}, out_path)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_rccl_backend_runs_on_this_box(tmp_path):
    """The multi-rank paths cannot run on a one-GPU box; what can be shown here is that torch's nccl backend (= RCCL)
    initialises on this GPU and carries the collectives those paths are built from (all_reduce, all_gather, broadcast,
    barrier) - so that a failure of the >= 2-GPU test below is a failure of the sharding logic, not of the environment."""
    import torch.multiprocessing as mp
    out = os.path.join(str(tmp_path), "single.pt")
    mp.spawn(_rccl_single_worker, args=(29653, out), nprocs=1, join=True)
    got = torch.load(out, weights_only=False)
    assert got["backend"] == "nccl"
    assert torch.equal(got["reduced"], torch.arange(8, dtype=torch.float32))
    assert torch.equal(got["gathered"], torch.arange(5)) and torch.equal(got["bcast"], torch.full((3,), 7.0))


@pytest.mark.timeout(900)
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank: this box has fewer than two")
def test_rccl_ranks_bitwise_equal_one_rank(tmp_path, ckpt_path, golden_dir):
    """BASELINE config 4's machinery on the real backend: min(8, #GPUs) ranks, backend nccl (= RCCL over xGMI), one GPU
    each.  all_gather of pooled vectors, the overlapped point-to-point gather in 4 pieces and the plain gather, several
    sequences as one job, the sharded F1-max / ROC area and the sharded pair list - every result bit-identical to the
    one-rank run (each score depends on its two graphs only)."""
    import torch.multiprocessing as mp
    from sg_pr_amd import synth, allpairs, sg_net, eval_batch
    from sg_pr_amd.parser_sg import sgpr_args
    world = min(8, torch.cuda.device_count())
    mp.spawn(_rccl_worker, args=(world, 29651, ckpt_path, golden_dir, str(tmp_path)), nprocs=world, join=True)
    got = torch.load(os.path.join(str(tmp_path), "rccl.pt"), weights_only=False)
    args = sgpr_args()
    args.model = ckpt_path
    trainer = sg_net.SGTrainer(args, False)
    scorer = allpairs.AllPairsScorer(model=trainer.model)
    mats = []
    for (seed, m), g_ in zip(((6, 403), (7, 131), (8, 7)), got["many"]):
        c, l, _, poses = synth.kitti_like_sequence(m, 100, seed)
        one = scorer.run(torch.from_numpy(c).cuda(), torch.from_numpy(l).cuda())
        assert torch.equal(one.cpu(), g_)
        mats.append((one, poses))
    assert torch.equal(mats[0][0].cpu(), got["full"])
    f1, auc = scorer.pr_roc(mats[0][0], mats[0][1])
    assert f1 == got["f1"] and auc == got["auc"]
    for r in range(world):                                               # the sharded (ungathered) row blocks
        blk = torch.load(os.path.join(str(tmp_path), "rows_r%d.pt" % r))
        assert torch.equal(blk["rows"], mats[1][0][blk["lo"]:blk["hi"]].cpu())
    data = os.path.join(golden_dir, "data")
    names = ["0.json", "3.json", "250.json"]
    pairs = [[os.path.join(data, a), os.path.join(data, b)] for a in names for b in names] * 3
    pred, gt = eval_batch.score_pair_list(trainer, pairs)
    np.testing.assert_array_equal(pred, got["pred"])
    np.testing.assert_array_equal(gt, got["gt"])


@pytest.mark.timeout(900)
def test_config4_full_size_on_one_gpu(oracle, oracle_sd, ckpt_path):
    """BASELINE config 4 at FULL size on one GPU: KITTI 00+02+05+06+08-sized sequences, 17 135 graphs, 67.75 M pairs.
    `allpairs.SequenceSet` (one embed launch for all graphs, one pair of launches for the five matrices) gives the
    matrices of per-sequence runs bit for bit; 2 500 sampled pairs (500 per sequence, over 1 000 distinct graphs each) are
    within 1e-4 of the oracle or involve a graph with a PROVEN kNN tie (tests/tie_proof.py); every matrix has the
    properties a correct one must have whatever its size (finite, in (0,1), asymmetric, pair-list kernel agreement on
    sampled pairs)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import tie_proof
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    from sg_pr_amd import synth, allpairs, sg_net
    from sg_pr_amd.parser_sg import sgpr_args
    frames = (("00", 4541), ("02", 4661), ("05", 2761), ("06", 1101), ("08", 4071))
    args = sgpr_args()
    args.model = ckpt_path
    model = sg_net.SGTrainer(args, False).model
    eng = model.engine()
    host, seqs = [], []
    for si, (_, m) in enumerate(frames):
        c, l, _, _ = synth.kitti_like_sequence(num_graphs=m, node_num=100, seed=si)
        host.append((c, l))
        seqs.append((torch.from_numpy(c).cuda(), torch.from_numpy(l).cuda()))
    assert sum(m for _, m in frames) == 17135 and sum(m * m for _, m in frames) == 67753965
    scorer = allpairs.AllPairsScorer(model=model)
    sset = allpairs.SequenceSet(scorer, seqs)
    order, cap = eng.size_order(sset.centers, sset.labels, 10)
    many = sset.run(embed_fn=lambda c, l: eng.embed(c, l, 10, node_cap=cap, order=order)[0])
    eng.check_status()
    rng = np.random.default_rng(4)
    worst, n_checked, n_excused = 0.0, 0, 0
    NS = 500                                                         # oracle-checked pairs per sequence
    for (name, m), (c, l), (dc, dl), got in zip(frames, host, seqs, many):
        assert got.shape == (m, m)
        one = scorer.run(dc, dl)                                     # the per-sequence job (eval_batch.py:26-36's loop body)
        assert torch.equal(one, got), name
        del one
        assert torch.isfinite(got).all() and float(got.min()) >= 0.0 and float(got.max()) <= 1.0
        assert not torch.equal(got[:64, :64], got[:64, :64].t())     # the NTN is asymmetric
        ii, jj = rng.integers(0, m, size=NS), rng.integers(0, m, size=NS)
        gi = np.concatenate((ii, jj))
        rp = torch.cat([oracle.embed(oracle_sd, torch.from_numpy(synth.dense_features(c[gi[s0:s0 + 250]], l[gi[s0:s0 + 250]])), 10)[0]
                        for s0 in range(0, 2 * NS, 250)])
        rs = oracle.score_from_pooled(oracle_sd, rp[:NS], rp[NS:])
        sample = got[torch.from_numpy(ii).cuda(), torch.from_numpy(jj).cuda()].cpu()
        d = (sample - rs).abs().numpy()
        pooled = eng.embed(dc[gi], dl[gi], 10)[0]
        dev = (pooled.cpu() - rp).abs().amax(1).numpy()
        pooled_h = pooled.cpu().numpy()
        for pi in np.flatnonzero(d > SCORE_TOL):                     # beyond the bar: only through a proven tie
            reps = []
            for q in (pi, NS + pi):
                if dev[q] > 1e-4:
                    rep = tie_proof.prove_graph(eng, oracle, oracle_sd, c[gi[q]], l[gi[q]], 10, pooled_h[q])
                    reps.append((int(gi[q]), rep["proven"], rep["reason"]))
            assert reps and any(ok for _, ok, _ in reps), "%s pair (%d, %d) differs by %.3g without a proven tie: %s" % (
                name, ii[pi], jj[pi], d[pi], reps)
            n_excused += 1
        clean = d <= SCORE_TOL
        worst = max(worst, float(d[clean].max()))
        n_checked += NS
        lst = eng.score_pairs(pooled[:NS].contiguous(), pooled[NS:].contiguous()).cpu()
        assert (lst - sample).abs().max().item() <= 5e-6             # pair-list kernel (fp32) vs the dense tail (f16 planes): 2.5e-6 over 2 500 pairs
    assert n_checked >= 2000 and n_excused <= n_checked // 100
    print("config 4 full size: %d oracle-checked pairs, worst max|dscore| within the bar = %.3e, pairs excused by a proven tie: %d"
          % (n_checked, worst, n_excused))


def _tail_float64(sd, rows, cols):
    """NTN + head (layers_batch.py:70-83, sg_net.py:131-136) for every (row, col) pair in float64 numpy."""
    w = sd["tensor_network.weight_matrix"].double().numpy()             # [32, 32, 16]
    wb = sd["tensor_network.weight_matrix_block"].double().numpy()      # [16, 64]
    bias = sd["tensor_network.bias"].double().numpy().reshape(-1)
    e1, e2 = rows.astype(np.float64), cols.astype(np.float64)
    bil = np.einsum("ri,ijt,cj->rct", e1, w, e2)
    blk = (e1 @ wb[:, :32].T)[:, None, :] + (e2 @ wb[:, 32:].T)[None, :, :]
    h = np.maximum(bil + blk + bias, 0.0)
    g = np.maximum(h @ sd["fully_connected_first.weight"].double().numpy().T + sd["fully_connected_first.bias"].double().numpy(), 0.0)
    z = g @ sd["scoring_layer.weight"].double().numpy().reshape(-1) + float(sd["scoring_layer.bias"])
    return 1.0 / (1.0 + np.exp(-z))


def test_all_pairs_f16_range_guard(eng, oracle_sd):
    """The dense tail forms f16 planes of A' = e1^T W + Wb, of the column vectors and of the hidden layer; inputs whose
    bound leaves the f16 range must take the exact fp32 path inside the same kernel (sgpr_score.hip) - results equal the
    pair-list kernel's (same arithmetic), including non-multiple-of-64 column counts and ragged row counts.  Accuracy
    of both kernels is held against a float64 evaluation: the two-plane f16 path is as close to it as the fp32 one."""
    rng = np.random.default_rng(11)
    rows_np = rng.normal(0, 4, size=(37, 32)).astype(np.float32)
    cols_np = rng.normal(0, 4, size=(131, 32)).astype(np.float32)
    rows, cols = torch.from_numpy(rows_np).cuda(), torch.from_numpy(cols_np).cuda()
    ii, jj = torch.meshgrid(torch.arange(37, dtype=torch.int32), torch.arange(131, dtype=torch.int32), indexing="ij")

    def both(scale):
        m = eng.score_all_pairs(rows * scale, cols * scale)
        lst = eng.score_pairs(rows * scale, cols * scale, ii.reshape(-1), jj.reshape(-1)).view(37, 131)
        return m.cpu().numpy(), lst.cpu().numpy()

    m, lst = both(1.0)                      # in range: the f16 two-plane MFMA path (|pooled| up to ~15: beyond real data)
    ref = _tail_float64(oracle_sd, rows_np, cols_np)
    ref_full = ref
    err_m, err_l = np.abs(m - ref).max(), np.abs(lst - ref).max()
    print("tail vs float64: all-pairs (f16 planes) %.3g, pair list (fp32) %.3g" % (err_m, err_l))
    assert err_m <= 3e-6 and err_l <= 3e-6
    assert 0.02 < m.mean() < 0.98           # not saturated: the comparison means something
    # the plane path has two forms, chosen per launch from a bound on |H| (sgpr_score.hip ap_mode): below 1024 the low
    # plane's ReLU rides on its conversion, above it is a packed maximum - 0.25 x is on the first side for certain, 2 x on
    # the second; both must be the same function
    for scale in (0.25, 2.0):
        m2, l2 = both(scale)
        ref2 = _tail_float64(oracle_sd, rows_np * np.float32(scale), cols_np * np.float32(scale))
        tol = 3e-6 * max(1.0, scale * scale)            # the hidden layer is bilinear in the inputs
        print("scale %g: all-pairs %.3g, pair list %.3g" % (scale, np.abs(m2 - ref2).max(), np.abs(l2 - ref2).max()))
        assert np.abs(m2 - ref2).max() <= tol and np.abs(l2 - ref2).max() <= tol, scale
    m, lst = both(400.0)                    # |A'| |e2| bound far beyond 65504: exact path, same arithmetic as the list
    np.testing.assert_array_equal(m, lst)
    m, lst = both(1e-4)                     # tiny inputs: subnormal lo planes
    ref = _tail_float64(oracle_sd, rows_np * np.float32(1e-4), cols_np * np.float32(1e-4))
    assert np.abs(m - ref).max() <= 1e-6 and np.abs(lst - ref).max() <= 1e-6
    # the instance at the reference's operand width (three bf16 planes per operand = 24 bits, fp32's range; debug bit 13 or
    # a checkpoint outside the f16 range selects it): <= 1e-6 of float64 where the values are O(1), the bilinear scaling of
    # the hidden layer beyond - and no range limit: what the f16 planes hand to the exact per-pair path it computes itself
    eng.set_skip_mask(8192)
    try:
        for scale, tol in ((1.0, 1e-6), (0.25, 1e-6), (2.0, 4e-6), (1e-4, 1e-6), (400.0, None)):
            mw = eng.score_all_pairs(rows * scale, cols * scale).cpu().numpy()
            refw = _tail_float64(oracle_sd, rows_np * np.float32(scale), cols_np * np.float32(scale))
            errw = np.abs(mw - refw).max()
            print("wide tail (3 x bf16 planes) vs float64 at scale %g: %.3g" % (scale, errw))
            if tol is not None:
                assert errw <= tol, (scale, errw)
            else:               # |pre-activations| ~ 1e7: scores saturate; the exact fp32 path is the yardstick there
                lstw = eng.score_pairs(rows * scale, cols * scale, ii.reshape(-1), jj.reshape(-1)).view(37, 131).cpu().numpy()
                assert np.isfinite(mw).all() and np.abs(mw - lstw).max() <= 1e-6
        # ragged edges: rows not a multiple of 16, columns not a multiple of 64 / 4
        for r_, c_ in ((1, 1), (17, 65), (37, 131), (16, 64)):
            mw = eng.score_all_pairs(rows[:r_].contiguous(), cols[:c_].contiguous()).cpu().numpy()
            assert np.abs(mw - ref_full[:r_, :c_]).max() <= 1e-6, (r_, c_)
    finally:
        eng.set_skip_mask(0)



def test_edges_of_the_tie_regime(eng, golden_dir):
    """Reference goldens for the graphs at the switch points of the kernel's branches: exactly k padded slots (one
    representative), a single label, < 17 processed slots (generic semantic branch), no padding with >= k nodes per
    label, trailing duplicate REAL nodes with and without padding, a one-node graph, k-1 padded slots with one label.
    Graph 0 (k-1 padded slots, mixed labels) is the counter-example: its padded rows have only k-1 zero-distance
    candidates and must take ONE real node, all of which tie at distance 1 with different features - torch.topk's
    choice there is implementation-defined, so it is held to determinism only."""
    g = np.load(os.path.join(golden_dir, "edge_n100_k10.npz"))
    pooled, att, emb = eng.embed(g["centers"], g["labels"], 10, want_att=True, want_emb=True)
    eng.check_status()
    defined = np.arange(1, 12)
    d_emb = np.abs(emb.cpu().numpy() - g["emb"]).reshape(12, -1).max(1)
    print("edge graphs: max|d emb| per graph", np.array2string(d_emb, precision=2))
    np.testing.assert_allclose(emb.cpu().numpy()[defined], g["emb"][defined], rtol=0, atol=EMB_TOL)
    np.testing.assert_allclose(att.cpu().numpy()[defined], g["att"][defined], rtol=0, atol=ATT_TOL)
    np.testing.assert_allclose(pooled.cpu().numpy()[defined], g["pooled"][defined], rtol=0, atol=1e-4)
    m = eng.score_all_pairs(pooled, pooled).cpu().numpy()
    err = np.abs(m - g["scores"])[np.ix_(defined, defined)].max()
    print("edge graphs: max|dscore| vs the reference =", err)
    assert err <= 2e-5
    # the same graphs through the capped / ordered launch and the dense entry point: bit-identical (graph 0 included)
    order, cap = eng.size_order(g["centers"], g["labels"], 10)
    p2, _, _ = eng.embed(g["centers"], g["labels"], 10, node_cap=cap, order=order)
    assert torch.equal(p2, pooled)
    from sg_pr_amd import synth
    p3, _, _ = eng.embed_dense(torch.from_numpy(synth.dense_features(g["centers"], g["labels"])), 10)
    assert torch.equal(p3, pooled)


def test_all_release_checkpoints(release_state_dicts, golden_dir):
    """Every checkpoint of the reference's release_model.zip through sgpr_create (BatchNorm fold in double, plane
    splits) and the kernels: the nine shipped pairs (dense drop-in forward) and a synthetic batch (packed path)."""
    from sg_pr_amd import engine
    g = np.load(os.path.join(golden_dir, "release_models.npz"))
    k3 = np.load(os.path.join(golden_dir, "kitti3_n100_k10.npz"))
    feats = torch.from_numpy(k3["features"])
    i, j = g["pair_ij"][:, 0].astype(np.int64), g["pair_ij"][:, 1].astype(np.int64)
    worst = 0.0
    for name, sd in release_state_dicts.items():
        e = engine.Engine(sd, device=0)
        try:
            assert e.uses_f16_planes(), name          # in range, and no row of tiny weights that the planes would starve
            s9, _, _ = e.forward_dense(feats[i], feats[j], 10)
            pooled, _, _ = e.embed(g["syn_centers"], g["syn_labels"], 10)
            ss = e.score_pairs(pooled[0::2].contiguous(), pooled[1::2].contiguous())
            d = max(np.abs(s9.cpu().numpy() - g["scores9/" + name]).max(),
                    np.abs(ss.cpu().numpy() - g["scores_syn/" + name]).max())
            assert d <= 2e-5, (name, d)
            worst = max(worst, d)
        finally:
            e.close()
    print("18 release checkpoints: worst max|dscore| =", worst)


def test_weights_outside_the_f16_range_take_the_wide_range_instance(oracle, ckpt_path):
    """A checkpoint whose folded weights exceed the f16 range is served by the wide-range instance throughout (decided at
    sgpr_create) and still follows the oracle; the 19 shipped checkpoints all run on the two-plane f16 datapath
    (test_all_release_checkpoints) although they hold whole channels of 1e-30 .. 1e-7 weights behind dead BatchNorm scales -
    the planes keep an absolute 2^-25 of those, which is all such a channel can matter."""
    from sg_pr_amd import engine, synth
    sd = {k: v.clone() for k, v in torch.load(ckpt_path, map_location="cpu").items()}
    key = [k for k in sd if k.endswith("dgcnn_s_conv2.0.weight")][0]
    pre = key[: -len("0.weight")]
    fold = sd[pre + "1.weight"] / torch.sqrt(sd[pre + "1.running_var"] + 1e-5)          # eval BatchNorm's scale per channel
    mag = fold.abs() * sd[key].flatten(1).abs().amax(1)
    ch = int(mag.argmax())                                # a live channel (the checkpoint has dead ones)
    sd[key][ch] *= 2e5 / float(mag[ch])                   # its folded weights now reach 2e5: beyond f16's 65504
    osd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}
    c, l, _ = synth.make_graphs(6, 100, 30, 60, 4, kitti_like=True)
    ref = oracle.embed(osd, torch.from_numpy(synth.dense_features(c, l)), 10)[0]
    e = engine.Engine(sd, device=0)
    try:
        assert not e.uses_f16_planes()
        p = e.embed(c, l, 10)[0].cpu()
    finally:
        e.close()
    assert torch.isfinite(p).all()
    assert ((p - ref).abs() <= 1e-4 + 1e-4 * ref.abs()).all()


def test_device_roc_auc(eng):
    """eval_batch.py:48-49 on the device: the ROC area from ranking every negative among the positive scores equals the
    sorted host computation (one counting pass, exact)."""
    from sg_pr_amd import synth, metrics, allpairs
    centers, labels, _, poses = synth.kitti_like_sequence(num_graphs=160, node_num=100, seed=8)
    pooled, _, _ = eng.embed(centers, labels, 10)
    m = eng.score_all_pairs(pooled, pooled)
    xz = allpairs.pose_xz(poses).cuda()
    auc = metrics.roc_auc_device(eng, m, pose_xz=xz)
    gt, valid = allpairs.ground_truth_mask(allpairs.pose_distance_matrix(poses), 3)
    want = metrics.roc_auc(gt[valid].numpy(), m.cpu()[valid].numpy())
    print("device AUC", auc, "host", want)
    assert abs(auc - want) < 1e-12
    # massive ties (scores rounded to two digits) and explicit labels with as many positives as negatives
    r = (m * 100).round() / 100
    g8 = (torch.rand(m.shape, generator=torch.Generator().manual_seed(1)) < 0.5).to(torch.int8)
    f1, auc, passes = metrics.pr_roc_device(eng, r, gt=g8)
    assert abs(auc - metrics.roc_auc(g8.numpy().ravel(), r.cpu().numpy().ravel())) < 1e-12
    assert abs(f1 - metrics.f1_max(g8.numpy().ravel(), r.cpu().numpy().ravel())) < 1e-12 and passes == 1


@pytest.mark.timeout(1200)
def test_config5_full_size(eng, oracle, oracle_sd):
    """BASELINE config 5 at FULL size: 1024 pairs = 2048 graphs, node_num 256, K 20 (the large LDS plan with the chunked
    key matrix).  ALL 1024 pairs against the oracle ("within 1e-4 or a proven kNN tie") plus size-independent properties
    over the whole batch."""
    from sg_pr_amd import synth
    centers, labels, _ = synth.config5_pairs(seed=0)
    assert centers.shape == (2048, 256, 3)
    order, cap = eng.size_order(centers, labels, 20)
    pooled, att, _ = eng.embed(centers, labels, 20, want_att=True, node_cap=cap, order=order)
    eng.check_status()
    scores = eng.score_pairs(pooled[0::2].contiguous(), pooled[1::2].contiguous())
    assert torch.isfinite(pooled).all() and torch.isfinite(scores).all()
    # ALL 1024 pairs against the oracle: every score within the 1e-4 bar, or one of the pair's graphs differs from the
    # oracle through a PROVEN kNN tie (tests/tie_proof.py: fp32-level gap in float64 on the oracle's own layer input, the
    # reference's own keys within a few ulp)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import tie_proof
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    rp_parts, ra_parts = [], []
    for s0 in range(0, 2048, 128):
        p_, a_, _ = oracle.embed(oracle_sd, torch.from_numpy(synth.dense_features(centers[s0:s0 + 128], labels[s0:s0 + 128])), 20)
        rp_parts.append(p_)
        ra_parts.append(a_.reshape(a_.shape[0], -1))
    rp, ra = torch.cat(rp_parts), torch.cat(ra_parts)
    rs = oracle.score_from_pooled(oracle_sd, rp[0::2], rp[1::2])
    d = (scores.cpu() - rs).abs().numpy()
    off = np.flatnonzero(d > SCORE_TOL)
    print("config 5 full size: max|dscore| over the 1024 pairs = %.3e; pairs beyond 1e-4: %d %s"
          % (d.max(), off.size, off.tolist()))
    assert off.size <= 10                                   # ties are rare events, not a regime
    pooled_h = pooled.cpu().numpy()
    dev = (pooled.cpu() - rp).abs().amax(1).numpy()
    excused = np.zeros(2048, dtype=bool)
    for pi in off:
        reps = []
        for g in (2 * pi, 2 * pi + 1):
            if dev[g] > 1e-4:
                rep = tie_proof.prove_graph(eng, oracle, oracle_sd, centers[g], labels[g], 20, pooled_h[g])
                reps.append((int(g), rep["proven"], rep["reason"], [round(f["ratio_fp32"], 3) for f in rep["flips"]]))
                excused[g] = rep["proven"]
        print("  pair %d (|d score| %.2e): %s" % (pi, d[pi], reps))
        assert reps and any(ok for _, ok, _, _ in reps), "pair %d differs by %.3g without a proven tie: %s" % (pi, d[pi], reps)
    # attention weights of the graphs whose embedding agrees (a graph with a flipped neighbour legitimately differs)
    # (|pooled| reaches ~25 at node_num 256: 3e-4 is this file's pooled gate for that shape)
    agree = dev <= 3e-4
    att_dev = np.abs(att.cpu().numpy() - ra.numpy()).max(1)
    print("config 5 full size: graphs with |d pooled| <= 3e-4: %d of 2048 (<= 1e-4: %d); max |d att| among them %.2e"
          % (agree.sum(), (dev <= 1e-4).sum(), att_dev[agree].max()))
    assert agree.sum() >= 2000 and att_dev[agree].max() <= 10 * ATT_TOL
    # plain launch (no cap, storage order) and a two-shard launch: bit-identical pooled vectors
    p_plain, _, _ = eng.embed(centers, labels, 20)
    assert torch.equal(p_plain, pooled)
    p_a, _, _ = eng.embed(centers[:1000], labels[:1000], 20)
    p_b, _, _ = eng.embed(centers[1000:], labels[1000:], 20)
    assert torch.equal(torch.cat((p_a, p_b)), pooled)
    # pair list == dense rectangle on a block
    blk = eng.score_all_pairs(pooled[:64], pooled[:256]).cpu()
    ii, jj = torch.meshgrid(torch.arange(64, dtype=torch.int32), torch.arange(256, dtype=torch.int32), indexing="ij")
    lst = eng.score_pairs(pooled, pooled, ii.reshape(-1), jj.reshape(-1)).view(64, 256).cpu()
    assert (blk - lst).abs().max().item() <= 2e-5      # |pooled| reaches ~25 here: both kernels are ~1e-5 from float64


def test_f16_planes_range_fallback(eng, oracle, oracle_sd):
    """embed_kernel keeps X as two f16 planes; a graph whose coordinates or activations reach the f16 range is flagged
    and embedded again by the wide-range instance (bf16 planes / fp32 rows) in the same call.  Huge coordinates (a
    scene in millimetres) must therefore still match the oracle, in the same batch as ordinary graphs, and the two
    layouts must agree on ordinary data."""
    from sg_pr_amd import synth
    centers, labels, _ = synth.make_graphs(24, 100, 25, 60, 31, kitti_like=True)
    big = centers.copy()
    big[::3] *= 2000.0                                        # every third graph: coordinates up to 1e5
    pooled, att, _ = eng.embed(big, labels, 10, want_att=True)
    eng.check_status()
    assert torch.isfinite(pooled).all()
    rp, ra, _ = oracle.embed(oracle_sd, torch.from_numpy(synth.dense_features(big, labels)), 10)
    s = eng.score_all_pairs(pooled, pooled).cpu()
    rs = oracle.score_all_pairs(oracle_sd, rp, rp)
    err = (s - rs).abs().max().item()
    print("mixed-range batch: max|dscore| vs oracle =", err)
    assert err <= SCORE_TOL
    # ordinary graphs are untouched by their neighbours' fallback: bit-identical to a batch without the huge ones
    p_small, _, _ = eng.embed(centers, labels, 10)
    keep = np.arange(24) % 3 != 0
    assert torch.equal(pooled[keep], p_small[keep])
    # the wide-range layout forced for every graph (debug mask bit 13) agrees with the f16 planes on ordinary data
    eng.set_skip_mask(8192)
    try:
        p_wide, _, _ = eng.embed(centers, labels, 10)
        order, cap = eng.size_order(centers, labels, 10)
        p_wide2, _, _ = eng.embed(centers, labels, 10, node_cap=cap, order=order)
    finally:
        eng.set_skip_mask(0)
    assert torch.equal(p_wide, p_wide2)
    d = (p_wide - p_small).abs().max().item()
    print("f16 planes vs bf16 planes: max|d pooled| =", d)
    assert d <= 1e-4
    # the flagged graphs went through the wide-range instance: same bits as forcing it
    eng.set_skip_mask(8192)
    try:
        p_big_wide, _, _ = eng.embed(big, labels, 10)
    finally:
        eng.set_skip_mask(0)
    assert torch.equal(p_big_wide[~keep], pooled[~keep])


@pytest.mark.parametrize("n,k,spread", [(128, 10, 50.0), (61, 10, 50.0), (256, 20, 100.0)])
def test_coordinate_layer_ranks_by_the_reference_fp32_keys(eng, oracle, n, k, spread):
    """The first xyz layer's neighbour sets are the reference's even between near-ties: the kernel restates the fp32
    expansion (dgcnn.py:14-20) operation for operation, so over ~60 k rows of random centres up to +-`spread` m no row
    may differ (two f16 planes flipped about one row in 10 k here).  Compared through the reference's own key values
    of the selected nodes, which is indifferent to the order among exactly equal keys."""
    rng = np.random.default_rng(77)
    g = 65536 // n
    centers = (rng.uniform(-spread, spread, (g, n, 3))).astype(np.float32)
    centers[:, :, 1] *= 0.1                                             # flat like a road scene: denser near-ties
    labels = rng.integers(0, 12, (g, n)).astype(np.int32)
    knn = eng.embed(centers, labels, k, debug=True)[4][:, 0].cpu().numpy().astype(np.int64)     # [G, N, k], layer xyz1
    pd = oracle.neg_sq_dist(torch.from_numpy(np.ascontiguousarray(centers.transpose(0, 2, 1))))
    ref = np.sort(pd.topk(k=k, dim=-1)[0].numpy(), -1)
    got = np.sort(np.take_along_axis(pd.numpy(), knn, -1), -1)
    bad = (ref != got).any(-1)
    assert not bad.any(), "rows with a neighbour set other than the reference's: %d of %d" % (bad.sum(), bad.size)
    assert (np.sort(knn, -1)[..., 1:] != np.sort(knn, -1)[..., :-1]).all()                       # k distinct nodes per row


@pytest.mark.timeout(1800)
def test_full_sequence_parity_with_tie_proofs(eng, oracle, oracle_sd):
    """BASELINE metric, second half ("F1-max parity") at FULL size on a sequence whose PR curve means something
    (synth.world_sequence: one world, revisits alike): all 4541 graphs through the engine and the oracle, both
    4541 x 4541 score matrices (eval_batch.py:30-36 restated as the dense job), F1-max of both (eval_batch.py:69-87).
    The north star's 1e-4 is held for every score between graphs whose embeddings agree; a graph whose embedding does
    not agree must come with a PROOF (tests/tie_proof.py) that it differs only through a kNN row whose two swapped
    candidates are closer, in float64 on the oracle's own layer input, than the reference's fp32 expansion
    (dgcnn.py:14-20) resolves - a selection bug or an upstream numerical error fails the proof, however few graphs it
    hits (tests/test_tie_proof.py holds the negative controls)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import tie_proof
    from sg_pr_amd import synth
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    c, l, _, poses = synth.world_sequence(4541, 100, seed=0)
    lines = []
    r = tie_proof.census(eng, oracle, oracle_sd, c, l, poses, log=lambda s: (print(s), lines.append(s)))
    assert not r["unproven"], "graphs that differ from the oracle without a proven tie: %s" % r["unproven"]
    assert r["clean_pooled_max"] < 1e-4
    assert r["scores"] == 4541 * 4541
    assert r["scores_off_clean"] == 0 and r["score_max_clean"] < SCORE_TOL       # every score between agreeing graphs
    assert r["flagged"].size <= 12                                                # ties are rare events, not a regime (observed: 3 - 6)
    # F1-max.  SURVEY 8d's |dF1| <= 1e-6 presumes score parity everywhere; it does not hold on the full matrix (observed
    # 3.6e-6): the proven-tie graphs move the 2 x 4541 scores of their rows and columns (by up to 5e-2), and scores that
    # agree to 2.5e-5 can still trade places across the best threshold.  Gate: 1e-5 on the full matrix (3 x the
    # observed) - the curve itself is far from chance (F1-max 0.146 against a positive rate of 0.2 %).
    assert r["f1_oracle"] > 0.1, "the sequence's PR curve is at chance: %r" % r["f1_oracle"]
    assert abs(r["f1_hip"] - r["f1_oracle"]) <= 1e-5
    assert abs(r["f1_hip_clean"] - r["f1_oracle_clean"]) <= 1e-5
    # every accepted flip fits the fp32 term of the bound ALONE (the capped input-rounding term was never needed)
    assert max(r.get("ratios_fp32") or [0.0]) <= 1.0, r.get("ratios_fp32")
    # the device F1-max (one engine call) on the HIP matrix = the sorted host computation on the same matrix
    pooled = r["pooled"]
    from sg_pr_amd import metrics
    dev_f1 = metrics.f1_max_device(eng, eng.score_all_pairs(pooled, pooled), pose_xz=np.ascontiguousarray(poses[:, [3, 11]]))[0]
    assert abs(dev_f1 - r["f1_hip"]) < 1e-12
    out = os.environ.get("SGPR_SEQ_PARITY_OUT")
    if out:
        with open(out, "w") as f:
            f.write("\n".join(lines) + "\n")


@pytest.mark.timeout(300)
def test_split_launch_late_producer_goes_to_the_second_pass(eng):
    """Split launch (<= 128 graphs: a graph's two branches on two workgroups): a consumer whose producer's flag does not
    arrive - HIP promises no co-residency between the workgroups of a launch - hands its graph to the second pass
    instead of hanging or failing.  Debug bit 20 makes the producers of the odd launch slots withhold their flag: the
    call takes the consumers' time-out (tens of milliseconds), reports no error, and every pooled vector is bit for bit the one of the
    undisturbed call."""
    from sg_pr_amd import synth
    c, l, _ = synth.make_graphs(24, 64, 20, 50, seed=21, kitti_like=True)        # node_num 64: the lean plan splits
    want, want_att, _ = eng.embed(c, l, 10, want_att=True)
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.embed(c, l, 10, want_att=True)
    torch.cuda.synchronize()
    t_plain = time.perf_counter() - t0
    eng.set_skip_mask(1 << 20)
    try:
        t0 = time.perf_counter()
        got, got_att, _ = eng.embed(c, l, 10, want_att=True)
        torch.cuda.synchronize()
        t_late = time.perf_counter() - t0
        eng.check_status()
    finally:
        eng.set_skip_mask(0)
    assert torch.equal(got, want) and torch.equal(got_att, want_att)
    # the hook really withheld flags: the call waited out the consumers' time-out (2^20 polls) instead of ~40 us
    assert t_late > 5e-3 and t_late > 20 * t_plain, (t_plain, t_late)
    again = eng.embed(c, l, 10)[0]                       # the request word of the disturbed launch does not linger
    assert torch.equal(again, want)


@pytest.mark.timeout(900)
def test_random_shapes_deviate_only_through_proven_ties(eng, oracle, oracle_sd):
    """The same statement away from the KITTI shape: 40 random (node_num 17..256, K 1..32, node counts) batches of 8
    graphs - the shapes of tools/exp/fuzz_oracle.py, whose one outlier (node_num 152, K 32: |d score| 2.2e-4) used to be a
    documented exception.  Every graph whose pooled vector moves by more than 2e-4 must be a proven kNN tie
    (tests/tie_proof.py); every score between the other graphs of its batch is within the 1e-4 bar."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import tie_proof
    from sg_pr_amd import synth
    rng = np.random.default_rng(2024)
    flagged_total, worst_clean, ratios = 0, 0.0, []
    for trial in range(40):
        n = int(rng.integers(17, 257))
        k = int(rng.integers(1, min(32, n // 2) + 1))
        hi = n - k
        lo = int(rng.integers(1, hi + 1))
        g = 8
        c, l, _ = synth.make_graphs(g, n, lo, hi, int(rng.integers(1 << 30)), kitti_like=bool(rng.integers(2)))
        ref = oracle.embed(oracle_sd, torch.from_numpy(synth.dense_features(c, l)), k)[0]
        pooled = eng.embed(c, l, k)[0]
        dev = (pooled.cpu() - ref).abs().amax(1).numpy()
        flagged = np.flatnonzero(dev > 2e-4)
        for gi in flagged:
            x_h, knn_h, p_dbg = tie_proof.hip_trace(eng, c[gi], l[gi], k)
            assert np.array_equal(p_dbg, pooled[gi].cpu().numpy())
            x_o, knn_o = tie_proof.oracle_trace(oracle, oracle_sd, torch.from_numpy(synth.dense_features(c[gi:gi + 1], l[gi:gi + 1])), k)
            rep = tie_proof.prove_ties(x_o, knn_o, x_h, knn_h)
            assert rep["proven"] and rep["flips"], (trial, n, k, int(gi), float(dev[gi]), rep["reason"])
            ratios += [f["ratio"] for f in rep["flips"]]
        flagged_total += flagged.size
        clean = np.setdiff1d(np.arange(g), flagged)
        if clean.size:
            ci = torch.from_numpy(clean)
            d = (eng.score_all_pairs(pooled[ci.cuda()].contiguous(), pooled[ci.cuda()].contiguous()).cpu() -
                 oracle.score_all_pairs(oracle_sd, ref[ci], ref[ci])).abs().max().item()
            worst_clean = max(worst_clean, d)
            assert d < SCORE_TOL, (trial, n, k, d)
    print("random shapes: %d of 320 graphs differ through proven ties (gap / bound up to %.2f); max |d score| between the "
          "others %.2e" % (flagged_total, max(ratios) if ratios else 0.0, worst_clean))
    assert flagged_total <= 8


def test_ragged_store_equals_padded_arrays(eng):
    """sgpr_embed_ragged (only the real nodes in memory, the padding made in registers) is bit-identical to the padded
    entry points: plain, with node_cap + launch order, with attention / embedding outputs, on graphs of 0 .. node_num
    nodes; a graph with more nodes than slots is a loud error."""
    from sg_pr_amd import synth
    from sg_pr_amd.engine import SgprError
    c, l, _, _ = synth.kitti_like_sequence(300, 100, seed=3)
    l = l.copy()
    c = c.copy()
    l[7], c[7] = -1, 0.0                                  # an empty graph
    l[9, :] = np.arange(100) % 12                         # a full one (no padding at all)
    c[9] = np.random.default_rng(0).uniform(-40, 40, (100, 3)).astype(np.float32)
    rc, rl, off = eng.to_ragged(c, l)
    assert rc.shape[0] == off[-1] == (l >= 0).sum() and off[8] == off[7] and off[10] - off[9] == 100
    ref_p, ref_a, ref_e = eng.embed(c, l, 10, want_att=True, want_emb=True)
    p1, a1, e1 = eng.embed_ragged(rc, rl, off, 100, 10, want_att=True, want_emb=True)
    assert torch.equal(p1, ref_p) and torch.equal(a1, ref_a) and torch.equal(e1, ref_e)
    order, cap = eng.ragged_order(off, 100, 10)
    order_p, cap_p = eng.size_order(c, l, 10)
    assert cap == cap_p == 100 and torch.equal(order, order_p)
    keep = np.setdiff1d(np.arange(300), [9])               # without the full graph the lean plan applies
    rc2, rl2, off2 = eng.to_ragged(c[keep], l[keep])
    order2, cap2 = eng.ragged_order(off2, 100, 10)
    assert cap2 <= 64
    p2 = eng.embed_ragged(rc2, rl2, off2, 100, 10, node_cap=cap2, order=order2)[0]
    assert torch.equal(p2, ref_p[torch.from_numpy(keep).cuda()])
    eng.check_status()
    bad = off.copy()
    bad[3:] += 120                                          # graph 2 now claims 120 + nodes: more than the 100 slots
    big_c = np.concatenate((rc, np.zeros((120, 3), np.float32)))
    big_l = np.concatenate((rl, np.zeros(120, np.int8)))
    pb = eng.embed_ragged(big_c, big_l, bad, 100, 10)[0]
    with pytest.raises(SgprError):
        eng.check_status()
    assert torch.isnan(pb[2]).all() and torch.equal(pb[:2], ref_p[:2])


def test_scorer_on_a_ragged_store_equals_padded_arrays(ckpt_path):
    """allpairs.RaggedGraphs through SG.embed, AllPairsScorer.run and SequenceSet.run: the matrices of the padded arrays
    bit for bit (the launch plan - node_cap, largest-first order - comes from the store's host-side offsets)."""
    from sg_pr_amd import allpairs, sg_net, synth
    from sg_pr_amd.parser_sg import sgpr_args
    args = sgpr_args()
    args.model = ckpt_path
    trainer = sg_net.SGTrainer(args, False)
    scorer = allpairs.AllPairsScorer(model=trainer.model)
    seqs = [synth.kitti_like_sequence(m, 100, seed)[:2] for seed, m in ((6, 203), (7, 77))]
    dev = torch.device("cuda", 0)
    padded = [(torch.from_numpy(c).to(dev), torch.from_numpy(l).to(dev)) for c, l in seqs]
    ragged = [(allpairs.RaggedGraphs.from_padded(c, l, device=dev), None) for c, l in seqs]
    ref = [scorer.run(c, l) for c, l in padded]
    got = [scorer.run(r, None) for r, _ in ragged]
    for a, b in zip(ref, got):
        assert torch.equal(a, b)
    many = allpairs.SequenceSet(scorer, ragged).run()
    for a, b in zip(ref, many):
        assert torch.equal(a, b)
    trainer.model.engine().check_status()


def test_generic_branch_graph_outside_f16_range(eng, oracle, oracle_sd):
    """The second pass (embed_redo_kernel) chains its two reasons: a graph that the lean plan hands over for the
    generic semantic branch (fewer than 17 processed slots) is embedded on the full f16 plan - and when THAT run leaves
    the f16 range (coordinates in millimetres) it must still reach the wide-range instance instead of keeping an
    overflowed pooled vector."""
    from sg_pr_amd import synth
    centers, labels, _ = synth.make_graphs(40, 100, 25, 60, 77, kitti_like=True)
    small = np.arange(0, 40, 4)                                # every fourth graph: 6 real nodes -> 7 processed slots
    centers[small, 6:] = 0.0
    labels[small, 6:] = -1
    big = centers.copy()
    big[small[::2]] *= 2000.0                                  # half of the small ones also leave the f16 range
    big[1::8] *= 2000.0                                        # ... and some ordinary graphs (flag 1 straight away)
    order, cap = eng.size_order(big, labels, 10)
    assert cap <= 64                                           # the lean plan (16-row park) is the one launched
    pooled, _, _ = eng.embed(big, labels, 10, node_cap=cap, order=order)
    eng.check_status()
    assert torch.isfinite(pooled).all()
    rp = oracle.embed(oracle_sd, torch.from_numpy(synth.dense_features(big, labels)), 10)[0]
    s = eng.score_all_pairs(pooled, pooled).cpu()
    rs = oracle.score_all_pairs(oracle_sd, rp, rp)
    err = (s - rs).abs().max().item()
    print("generic-branch graphs outside the f16 range: max|dscore| vs oracle =", err)
    assert err <= SCORE_TOL
    eng.set_skip_mask(8192)                                    # every graph on the wide-range instance
    try:
        p_wide, _, _ = eng.embed(big, labels, 10)
    finally:
        eng.set_skip_mask(0)
    hit = np.zeros(40, dtype=bool)
    hit[small[::2]] = True
    hit[1::8] = True
    assert torch.equal(p_wide[hit], pooled[hit])               # the out-of-range graphs: same bits as forcing the wide instance
    p_plain, _, _ = eng.embed(big, labels, 10)                 # uncapped plan (no second-pass reason 2): same results
    assert torch.equal(p_plain, pooled)


def test_f1_max_one_call(eng):
    """sgpr_f1_max (positives, thresholds, counting passes and the F1 reduction in one engine call, one 64-byte copy back)
    equals the sorted host computation to < 1e-12 - poses and explicit labels, ties, ragged shapes, no positives, only
    positives - and reports the rectangles it is not built for, which metrics.f1_max_device then settles on the
    multi-call path."""
    from sg_pr_amd import synth, metrics, allpairs
    centers, labels, _, poses = synth.kitti_like_sequence(num_graphs=900, node_num=100, seed=5)
    order, cap = eng.size_order(centers, labels, 10)
    pooled = eng.embed(centers, labels, 10, node_cap=cap, order=order)[0]
    m = eng.score_all_pairs(pooled, pooled)
    xz = allpairs.pose_xz(poses)
    d = torch.cdist(xz.double(), xz.double())
    gt = torch.where(d <= 3, 1, torch.where(d >= 20, 0, -1)).to(torch.int8)

    def host(sc, lab):
        keep = lab.ravel() >= 0
        return metrics.f1_max(lab.ravel()[keep], sc.ravel()[keep])

    res = eng.f1_max(m, pose_xz=xz)
    want = host(m.cpu().numpy(), gt.numpy())
    print("one-call F1-max", res.tolist(), "host", want)
    assert res[1] == 0 and abs(res[0] - want) < 1e-12
    assert res[2] == int((gt == 1).sum()) and res[3] == int((gt == 0).sum()) and res[4] in (1, 2)
    assert metrics.f1_max_device(eng, m, pose_xz=xz) == metrics.f1_max_device(eng, m, pose_xz=xz, one_call=False)
    # a row shard with explicit labels; massive ties; a padded leading dimension; tiny shapes
    res = eng.f1_max(m[101:358], row0=101, gt=gt[101:358])
    assert res[1] == 0 and abs(res[0] - host(m[101:358].cpu().numpy(), gt[101:358].numpy())) < 1e-12
    r2 = (m * 50).round() / 50
    res = eng.f1_max(r2, pose_xz=xz)
    assert res[1] == 0 and abs(res[0] - host(r2.cpu().numpy(), gt.numpy())) < 1e-12
    rng = np.random.default_rng(12)
    for rows, cols, ld, ppos in ((37, 333, 340, 0.3), (1, 5, 5, 0.5), (129, 1023, 1023, 0.02), (3, 2, 8, 0.5), (300, 300, 300, 0.0),
                                 (64, 64, 64, 1.0)):
        buf = torch.rand(rows, ld, generator=torch.Generator().manual_seed(rows + cols)).cuda()
        sc = buf[:, :cols]
        lab = np.where(rng.random((rows, cols)) < 0.15, -1, (rng.random((rows, cols)) < ppos).astype(np.int8)).astype(np.int8)
        res = eng.f1_max(sc, gt=torch.from_numpy(lab))
        assert res[1] == 0, (rows, cols, res.tolist())
        assert abs(res[0] - host(sc.cpu().numpy(), lab)) < 1e-12, (rows, cols, res.tolist())
        assert res[2] == int((lab == 1).sum()) and res[3] == int((lab == 0).sum())
    # as many positives as negatives, random scores: the F1 curve is flat, far more than 8191 values stay open after
    # the first pass - the engine says so (status 1) and the multi-call path settles it
    big = torch.rand(1100, 1100, generator=torch.Generator().manual_seed(2)).cuda()
    lab = (rng.random((1100, 1100)) < 0.5).astype(np.int8)
    res = eng.f1_max(big, gt=torch.from_numpy(lab))
    f1, passes = metrics.f1_max_device(eng, big, gt=torch.from_numpy(lab))
    print("flat curve: one-call status", res[1], "open values", res[6], "-> multi-call passes", passes)
    assert abs(f1 - host(big.cpu().numpy(), lab)) < 1e-12
    assert res[1] in (0, 1) and (res[1] == 1 or abs(res[0] - f1) < 1e-12)


def test_f1_max_strips_culled_by_bounding_box(eng):
    """sgpr_f1_max in pose mode calls whole (4 rows x 256 columns) blocks negative from fp32 bounding boxes of the columns'
    poses, without the reference's float64 arithmetic per pair (utils.py:36) - so the classes it implies must be the
    reference's wherever a pair is near a threshold or a pose is degenerate: trajectories far from the origin (fp32
    rounding of the coordinates: metres at 1e9), frames exactly d_neg apart, clusters where nothing is far, NaN / infinite
    poses, other thresholds, ragged sizes, a row shard.  Counts of positives / negatives and F1-max against the host."""
    from sg_pr_amd import metrics
    rng = np.random.default_rng(321)

    def classes(xz, r0, r1, d_pos, d_neg):
        dx = xz[r0:r1, None, 0] - xz[None, :, 0]
        dz = xz[r0:r1, None, 1] - xz[None, :, 1]
        with np.errstate(invalid="ignore", over="ignore"):
            d = np.sqrt(dx * dx + dz * dz)
            return np.where(d <= d_pos, 1, np.where(d >= d_neg, 0, -1)).astype(np.int8)

    def check(xz, what, d_pos=3.0, d_neg=20.0, rows=None):
        n = xz.shape[0]
        r0, r1 = rows if rows else (0, n)
        sc = torch.rand(r1 - r0, n, generator=torch.Generator().manual_seed(n + r0)) ** 2
        lab = classes(xz, r0, r1, d_pos, d_neg)
        res = eng.f1_max(sc.cuda(), row0=r0, pose_xz=torch.from_numpy(xz), d_pos=d_pos, d_neg=d_neg)
        assert res[2] == int((lab == 1).sum()) and res[3] == int((lab == 0).sum()), (what, res.tolist(), int((lab == 1).sum()), int((lab == 0).sum()))
        if res[1] == 0:
            keep = lab.ravel() >= 0
            want = metrics.f1_max(lab.ravel()[keep], sc.numpy().ravel()[keep]) if keep.any() else 0.0
            assert abs(res[0] - want) < 1e-12, (what, res.tolist(), want)
        else:
            assert res[1] == 1, (what, res.tolist())       # (a flat curve: the multi-call path's business)

    def walk(n, step=1.2, turn=0.15):
        heading = np.cumsum(rng.normal(0.0, turn, size=n))
        return np.cumsum(np.stack([np.cos(heading), np.sin(heading)], axis=1) * step, axis=0)

    base = walk(1500)
    base[-400:] = base[200:600] + rng.normal(0.0, 0.4, size=(400, 2))       # a revisit
    check(base, "trajectory")
    check(base[:777], "ragged size")
    check(base, "row shard", rows=(301, 655))
    for off in (1e4, 1e6, 1e9):
        check(base + np.array([off, -0.37 * off]), "far from the origin %g" % off)
    check(base, "thresholds 0.5 / 2", d_pos=0.5, d_neg=2.0)
    check(base, "thresholds 50 / 400", d_pos=50.0, d_neg=400.0)
    check(base, "d_neg below d_pos", d_pos=20.0, d_neg=3.0)
    grid = np.stack(np.meshgrid(np.arange(40) * 20.0, np.arange(30) * 20.0), axis=-1).reshape(-1, 2).astype(np.float64)
    check(grid, "frames exactly d_neg apart")
    check(grid * (1.0 - 1e-9), "a hair closer than d_neg")
    check(grid + 3e5, "the same grid far from the origin")
    check(rng.normal(0.0, 4.0, size=(600, 2)), "one cluster")
    line = np.stack([np.arange(1200) * 0.05, np.zeros(1200)], axis=1)
    check(line, "a slow straight line")
    bad = base.copy()
    bad[[5, 300, 301, 1499]] = np.nan
    bad[[17, 900]] = np.inf
    bad[600, 0] = -np.inf
    check(bad, "NaN and infinite poses")


def test_bench_on_real_graph_directory(golden_dir):
    """bench.py --data-dir packs a directory of real graph JSONs once (graph_store.pack_directory) and benchmarks their
    all-pairs matrix: the three shipped graphs here, $SG_PR_DATA/graphs_sk/00 on a machine that has KITTI mounted."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--data-dir", os.path.join(golden_dir, "data"),
                          "--steps", "3", "--warmup", "1", "--prewarm", "0", "--no-cpu-baseline"],
                         check=True, capture_output=True, text=True, timeout=300).stdout
    line = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    assert line["data"] == "real" and line["config"]["graphs"] == 3 and line["config"]["pairs_per_step"] == 9
    assert line["value"] > 0 and line["end_to_end"]["d2h"]["value"] > 0


def test_f16_planes_against_wide_range_on_the_shipped_graphs(eng, golden_dir):
    """the default datapath keeps operands as two f16 planes (22 bits); the wide-range instance (three bf16
    planes, 24 bits) is the reference point for what that costs on REAL graphs.  On the three shipped KITTI graphs the two
    must give the same neighbour sets (the f16 path's lists equal the golden's in test_shipped_graphs_every_intermediate;
    here: the pooled vectors, attention weights and all nine scores of the two datapaths agree far inside the gates)."""
    g = np.load(os.path.join(golden_dir, "kitti3_n100_k10.npz"))
    centers, labels = _packed_from_golden_features(g["features"])
    p16, a16, _ = eng.embed(centers, labels, 10, want_att=True)
    eng.set_skip_mask(8192)                                    # every graph on the wide-range instance
    try:
        pw, aw, _ = eng.embed(centers, labels, 10, want_att=True)
    finally:
        eng.set_skip_mask(0)
    dp = (p16 - pw).abs().max().item()
    da = (a16 - aw).abs().max().item()
    s16 = eng.score_all_pairs(p16, p16).cpu().numpy()
    eng.set_skip_mask(8192)                                    # ... and the tail's (three bf16 planes per operand)
    try:
        sw = eng.score_all_pairs(pw, pw).cpu().numpy()
    finally:
        eng.set_skip_mask(0)
    ds = float(np.abs(s16 - sw).max())
    print("f16 planes vs wide range on the shipped graphs: max|d pooled| %.3g, max|d att| %.3g, max|d score| %.3g" % (dp, da, ds))
    assert dp <= 2e-5 and da <= 2e-6 and ds <= 5e-6
    np.testing.assert_allclose(s16.reshape(-1), g["scores"], rtol=0, atol=GOLDEN_SCORE_TOL)
    np.testing.assert_allclose(sw.reshape(-1), g["scores"], rtol=0, atol=GOLDEN_SCORE_TOL)


def _listed(eng, pooled_r, pooled_c, i1, i2):
    plan = eng.pair_plan(i1, i2, pooled_r.shape[0], pooled_c.shape[0])
    return eng.score_pair_list(pooled_r, pooled_c, plan), plan


def test_grouped_pair_list_is_bitwise_the_dense_matrix(eng):
    """sgpr_score_pair_list (the reference's loop shape, eval_batch.py:30-36: a pair LIST grouped by row graph) against
    the dense rectangle: every listed score is bit for bit the matrix entry at (row, column), for rows with 1, 15, 16,
    17 and 40 listed pairs, absent rows, repeated pairs, a rectangular job and a one-pair list; the one-wave-per-pair
    kernel (exact fp32) agrees to rounding."""
    from sg_pr_amd import synth
    c, l, _ = synth.make_graphs(300, 100, 25, 60, seed=5, kitti_like=True)
    pooled = eng.embed(c, l, 10)[0]
    dense = eng.score_all_pairs(pooled, pooled)
    rng = np.random.default_rng(11)
    i1, i2 = [], []
    for row, cnt in ((0, 1), (3, 15), (4, 16), (7, 17), (299, 40), (150, 33), (151, 2)):
        i1 += [row] * cnt
        i2 += rng.integers(0, 300, cnt).tolist()
    i1 += [7, 7, 7]
    i2 += [5, 5, 5]                                             # repeated pairs
    extra = rng.integers(0, 300, (4000, 2))
    i1, i2 = np.array(i1 + extra[:, 0].tolist()), np.array(i2 + extra[:, 1].tolist())
    perm = rng.permutation(i1.size)
    i1, i2 = i1[perm], i2[perm]
    got, plan = _listed(eng, pooled, pooled, i1, i2)
    assert plan.P == i1.size and plan.n_rows == np.unique(i1).size
    want = dense[torch.from_numpy(i1).cuda(), torch.from_numpy(i2).cuda()]
    assert torch.equal(got, want)
    one_wave = eng.score_pairs(pooled, pooled, torch.from_numpy(i1.astype(np.int32)), torch.from_numpy(i2.astype(np.int32)))
    assert (got - one_wave).abs().max().item() < 3e-6
    # rectangular: rows and columns from different graph sets
    rect = eng.score_all_pairs(pooled[:37].contiguous(), pooled[100:].contiguous())
    j1, j2 = rng.integers(0, 37, 777), rng.integers(0, 200, 777)
    got2, _ = _listed(eng, pooled[:37].contiguous(), pooled[100:].contiguous(), j1, j2)
    assert torch.equal(got2, rect[torch.from_numpy(j1).cuda(), torch.from_numpy(j2).cuda()])
    one, _ = _listed(eng, pooled, pooled, [299], [0])
    assert torch.equal(one, dense[299, 0:1])
    empty, _ = _listed(eng, pooled, pooled, [], [])
    assert empty.numel() == 0
    # inputs outside the f16 range take the exact fp32 path inside the same kernel - like the dense kernel does
    big = pooled * 400.0
    dense_big = eng.score_all_pairs(big, big)
    got_big, _ = _listed(eng, big, big, i1, i2)
    assert torch.equal(got_big, dense_big[torch.from_numpy(i1).cuda(), torch.from_numpy(i2).cuda()])
    # errors: an index outside the graphs, a plan for other sizes
    from sg_pr_amd.engine import SgprError
    with pytest.raises(SgprError):
        eng.pair_plan([0, 300], [0, 0], 300, 300)
    with pytest.raises(ValueError):
        eng.score_pair_list(pooled[:10].contiguous(), pooled, plan)


@pytest.mark.timeout(900)
def test_reference_pair_lists_full_size(eng, golden_dir, ckpt_path, oracle, oracle_sd):
    """The reference's own evaluation lists (data_process/pair_list/pair_list_3_20_{02,05,06,08}.npy, index pairs kept as
    tests/golden/pair_lists_3_20.npz) at full size over KITTI-like graphs: grouped scores == dense matrix entries bit
    for bit, and SG.score_pooled routes lists of this size through the grouped kernel."""
    from sg_pr_amd import synth, sg_net
    from sg_pr_amd.parser_sg import sgpr_args
    fx = np.load(os.path.join(golden_dir, "pair_lists_3_20.npz"))
    args = sgpr_args()
    args.model = ckpt_path
    model = sg_net.SGTrainer(args, False).model
    for si, name in enumerate(("06", "02")):
        ij = fx["seq_" + name].astype(np.int64)
        m = int(ij.max()) + 1
        assert (name, m, ij.shape[0]) in (("06", 1101, 8299), ("02", 4661, 71226))
        c, l, _, _ = synth.kitti_like_sequence(m, 100, seed=20 + si)
        pooled = eng.embed(c, l, 10)[0]
        perm = np.random.default_rng(si).permutation(ij.shape[0])
        i1, i2 = ij[perm, 0], ij[perm, 1]
        got, plan = _listed(eng, pooled, pooled, i1, i2)
        assert plan.n_rows == np.unique(i1).size and plan.n_items >= plan.n_rows
        dense = eng.score_all_pairs(pooled, pooled)
        assert torch.equal(got, dense[torch.from_numpy(i1).cuda(), torch.from_numpy(i2).cuda()])
        via_model = model.score_pooled(pooled, pooled, torch.from_numpy(i1.astype(np.int32)), torch.from_numpy(i2.astype(np.int32)))
        assert torch.equal(via_model, got)
        if name == "02":
            # ... and the ORACLE on a sample of the list (not only HIP against HIP): the listed pairs of the first row
            # graphs until 2000 pairs are reached, both graphs of every pair embedded by the oracle, the reference's
            # per-pair tail on them (eval_batch.py:30-36 restated).  A graph whose embedding differs must be a proven
            # kNN tie (tests/tie_proof.py); every score between the others is within the 1e-4 bar.
            order = np.argsort(ij[:, 0], kind="stable")
            take = order[:np.searchsorted(ij[order, 0], ij[order[2000], 0], side="right")]
            assert take.size >= 2000
            used = np.unique(ij[take])
            remap = np.full(m, -1, dtype=np.int64)
            remap[used] = np.arange(used.size)
            torch.set_num_threads(min(64, os.cpu_count() or 8))
            rp = torch.cat([oracle.embed(oracle_sd, torch.from_numpy(synth.dense_features(c[used[s0:s0 + 256]], l[used[s0:s0 + 256]])), 10)[0]
                            for s0 in range(0, used.size, 256)])
            rs = oracle.score_from_pooled(oracle_sd, rp[remap[ij[take, 0]]], rp[remap[ij[take, 1]]]).numpy()
            inv = np.empty_like(perm)
            inv[perm] = np.arange(perm.size)
            hs = got.cpu().numpy()[inv[take]]                    # got is in the shuffled list's order
            dev = (pooled.cpu()[torch.from_numpy(used)] - rp).abs().amax(1).numpy()
            flagged = used[dev > 2e-4]
            import sys
            sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
            import tie_proof
            ph = pooled.cpu().numpy()
            for g in flagged:
                rep = tie_proof.prove_graph(eng, oracle, oracle_sd, c[g], l[g], 10, ph[g])
                assert rep["proven"], (int(g), rep["reason"])
            touched = np.isin(ij[take, 0], flagged) | np.isin(ij[take, 1], flagged)
            dd = np.abs(hs - rs)
            print("pair list 02: %d listed pairs over %d graphs against the oracle: max |d score| %.3e between graphs whose "
                  "embeddings agree (%d pairs touch the %d proven-tie graphs, max |d| there %.3e)"
                  % (take.size, used.size, dd[~touched].max(), int(touched.sum()), flagged.size,
                     dd[touched].max() if touched.any() else 0.0))
            assert dd[~touched].max() <= SCORE_TOL and flagged.size <= max(1, used.size // 100)


def test_f1_max_bit_pattern_bins_at_their_edges(eng):
    """sgpr_f1_max bins the negatives by the score's bit pattern (f1_key: 5 mantissa bits of s below 1/2, of 1 - s above,
    of s again beyond 1).  Scores ON the seams of that map - 0, subnormals, 1/2, 1, the neighbours of each, values beyond
    1, +inf - and whole matrices squeezed into one or two bins must still give the sorted host computation exactly."""
    from sg_pr_amd import metrics
    rng = np.random.default_rng(99)
    one = np.float32(1.0)
    special = np.array([0.0, 1e-45, 1e-40, 1.1754944e-38, 1e-20, 0.25, np.nextafter(np.float32(0.5), np.float32(0)), 0.5,
                        np.nextafter(np.float32(0.5), one), 0.75, 1 - 2.0 ** -7, 1 - 2.0 ** -20, np.nextafter(one, np.float32(0)),
                        1.0, np.nextafter(one, np.float32(2)), 1.5, 2.0, 1e10, 3e38, np.inf], dtype=np.float32)

    def check(sc, lab, what, may_fall_back=False):
        dev = torch.from_numpy(sc).cuda()
        res = eng.f1_max(dev, gt=torch.from_numpy(lab))
        keep = lab.ravel() >= 0
        want = metrics.f1_max(lab.ravel()[keep], sc.ravel()[keep])
        # status 1 = "more than 4095 distinct positive scores left to settle": legitimate when thousands of distinct
        # positive values share a few bins; metrics.f1_max_device then takes the multi-call path
        assert res[1] == 0 or (may_fall_back and res[1] == 1), (what, res.tolist())
        if res[1] == 0:
            assert abs(res[0] - want) < 1e-12, (what, res.tolist(), want)
            assert res[2] == int((lab == 1).sum()) and res[3] == int((lab == 0).sum()), (what, res.tolist())
        assert abs(metrics.f1_max_device(eng, dev, gt=torch.from_numpy(lab))[0] - want) < 1e-12, what

    for rows, cols in ((64, 257), (200, 200)):
        lab = rng.integers(-1, 2, size=(rows, cols)).astype(np.int8)
        check(special[rng.integers(0, special.size, size=(rows, cols))], lab, "special values")
        # a sigmoid's crowd near 1: 1 - 10^-u, u uniform in [0, 7]; positives a little higher than negatives
        u = rng.uniform(0, 7, size=(rows, cols)) + 0.3 * (lab == 1)
        check((1.0 - 10.0 ** -u).astype(np.float32), lab, "crowd near 1")
        # everything inside ONE bin (64 consecutive bit patterns share the top bits), and inside two neighbours
        base = np.float32(0.8125).view(np.uint32)
        check((base + rng.integers(0, 64, size=(rows, cols)).astype(np.uint32)).view(np.float32), lab, "one bin")
        check((base + rng.integers(0, 2 ** 17 + 64, size=(rows, cols)).astype(np.uint32)).view(np.float32), lab, "few bins",
              may_fall_back=True)
    # rare positives with scores spread over the whole range among 10^6 negatives (the loop-closure shape)
    sc = rng.random((1000, 1000), dtype=np.float32) ** 3
    lab = (rng.random((1000, 1000)) < 0.002).astype(np.int8)
    sc[lab == 1] = np.sqrt(sc[lab == 1])
    check(sc, lab, "rare positives")


def test_smaller_architectures_run_on_the_built_kernels(oracle):
    """parser_sg.py:12-18 exposes filters_1/2/3, tensor_neurons and bottle_neck_neurons; the label count is a model
    argument (sg_net.py:40-76).  Every architecture no larger than the built one {12, 64, 64, 32, 16, 16} in any of the six
    is served by zero-padding its tensors at sgpr_create - exact, not approximate: randomly initialised models of three
    such shapes (BatchNorm statistics randomised too) against the oracle, through the reference's own SG / state-dict API;
    a larger one is refused with SGPR_E_DIMS."""
    from sg_pr_amd import sg_net, synth
    from sg_pr_amd.engine import SgprError
    from sg_pr_amd.parser_sg import sgpr_args
    rng = np.random.default_rng(5)
    for labels, f1, f2, f3, t, bn in ((10, 32, 48, 16, 8, 8), (12, 64, 64, 32, 16, 16), (5, 17, 33, 9, 3, 5), (12, 64, 64, 32, 16, 1)):
        args = sgpr_args()
        args.filters_1, args.filters_2, args.filters_3, args.tensor_neurons, args.bottle_neck_neurons = f1, f2, f3, t, bn
        args.node_num, args.K = 64, 10
        torch.manual_seed(labels * 1000 + f1)
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
                if name.endswith(".1.bias") or name in ("fully_connected_first.bias", "scoring_layer.bias"):
                    prm.copy_(torch.randn_like(prm) * 0.2)
        model.eval()
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        c, l, _ = synth.make_graphs(24, 64, 20, 50, seed=labels)
        l = np.where(l >= 0, l % labels, l).astype(np.int32)
        for gi in range(l.shape[0]):                                  # keep the label-ascending node order of the generators
            n = int((l[gi] >= 0).sum())
            order = np.argsort(l[gi, :n], kind="stable")
            l[gi, :n], c[gi, :n] = l[gi, :n][order], c[gi, :n][order]
        feats = torch.from_numpy(synth.dense_features(c, l, num_labels=labels))
        f_a, f_b = feats[:12], feats[12:]
        want, wa1, wa2 = oracle.forward(sd, f_a, f_b, 10)
        got, a1, a2 = model({"features_1": f_a, "features_2": f_b})
        assert got.shape == (12,) and a1.shape == (12, 64, 1)
        assert (got.cpu() - want).abs().max().item() < SCORE_TOL, (labels, f1, f2, f3, t, bn)
        assert (a1.cpu() - wa1).abs().max().item() < 1e-4 and (a2.cpu() - wa2).abs().max().item() < 1e-4
        emb = model.dgcnn_conv_pass(feats)
        assert emb.shape == (24, 64, f3)
        assert (emb.cpu() - oracle.conv_pass(sd, feats, 10)).abs().max().item() < 1e-4
        # packed input and the all-pairs tail
        pooled = model.embed(c, l)[0]
        assert pooled.shape == (24, 32) and (f3 == 32 or float(pooled[:, f3:].abs().max()) == 0.0)
        ref_pooled = oracle.embed(sd, feats, 10)[0]
        assert (pooled[:, :f3].cpu() - ref_pooled).abs().max().item() < 2e-4
        mat = model.score_all_pairs(pooled, pooled).cpu()
        assert (mat - oracle.score_all_pairs(sd, ref_pooled, ref_pooled)).abs().max().item() < SCORE_TOL
        # a label the model does not have is an error, as in the reference (KeyError, sg_net.py:277)
        if labels < 12:
            bad = l.copy()
            bad[0, 0] = labels
            model.embed(torch.from_numpy(c).cuda(), torch.from_numpy(bad).cuda())
            with pytest.raises(SgprError):
                model.engine().check_status()
        # the stand-alone modules with this model's widths
        from sg_pr_amd.layers_batch import AttentionModule, TenorNetworkModule
        att_mod = AttentionModule(args).cuda()
        att_mod.load_state_dict({"weight_matrix": sd["attention.weight_matrix"]})
        e = torch.randn(3, 20, f3).cuda()
        rep, sig = att_mod(e)
        w_rep, w_sig = oracle.attention({"attention.weight_matrix": sd["attention.weight_matrix"]}, e.cpu())
        assert rep.shape == (3, f3, 1) and (rep.cpu() - w_rep).abs().max().item() < 1e-4
        ntn_mod = TenorNetworkModule(args).cuda()
        ntn_mod.load_state_dict({k.split(".", 1)[1]: v for k, v in sd.items() if k.startswith("tensor_network.")})
        e1, e2 = torch.randn(7, f3, 1).cuda(), torch.randn(7, f3, 1).cuda()
        ntn = ntn_mod(e1, e2)
        assert ntn.shape == (7, t, 1) and (ntn.cpu() - oracle.tensor_network(sd, e1.cpu(), e2.cpu())).abs().max().item() < 1e-4
    args = sgpr_args()
    args.filters_3 = 129                                              # beyond what the any-shape kernels serve
    big = sg_net.SG(args, 12).eval()
    with pytest.raises(SgprError, match="SGPR_E_DIMS"):
        big.engine()


def _randomised(model):
    """a randomly initialised SG with BatchNorm statistics, scales and the biases randomised too -> (model, state dict)"""
    with torch.no_grad():
        for name, buf in model.named_buffers():
            if name.endswith("running_mean"):
                buf.copy_(torch.randn_like(buf) * 0.2)
            if name.endswith("running_var"):
                buf.copy_(torch.rand_like(buf) + 0.5)
        for name, prm in model.named_parameters():
            if name.endswith(".1.weight"):
                prm.copy_(torch.rand_like(prm) + 0.5)
            if name.endswith(".1.bias") or name in ("fully_connected_first.bias", "scoring_layer.bias"):
                prm.copy_(torch.randn_like(prm) * 0.2)
    model.eval()
    return model, {k: v.detach().clone() for k, v in model.state_dict().items()}


def _label_sorted_graphs(count, node_num, lo, hi, labels, seed):
    from sg_pr_amd import synth
    c, l, _ = synth.make_graphs(count, node_num, lo, hi, seed=seed)
    spread = (l * 5 + 3 + np.arange(l.shape[0])[:, None]) if labels > 12 else l          # (reach the label channels past 12)
    l = np.where(l >= 0, spread % labels, l).astype(np.int32)
    for gi in range(l.shape[0]):                                      # the label-ascending node order of the generators
        n = int((l[gi] >= 0).sum())
        order = np.argsort(l[gi, :n], kind="stable")
        l[gi, :n], c[gi, :n] = l[gi, :n][order], c[gi, :n][order]
    return c, l


def test_larger_architectures_run_on_the_any_shape_kernels(oracle):
    """parser_sg.py:12-22 takes ANY filters_1/2/3, tensor_neurons, bottle_neck_neurons, node_num and K; sg_net.py:40-76 any
    label count.  Architectures larger than the built shape {12, 64, 64, 32, 16, 16} get an any-shape handle (plain-fp32
    kernels, sgpr_generic.hip): every entry point of the path against the oracle, through the reference's own SG API -
    dense forward, conv pass, packed / ragged embed, list and all-pairs tails, the bad-label report."""
    from sg_pr_amd import sg_net, synth
    from sg_pr_amd.allpairs import RaggedGraphs
    from sg_pr_amd.engine import SgprError
    from sg_pr_amd.parser_sg import sgpr_args
    for labels, f1, f2, f3, t, bn, node_num, K in ((12, 96, 128, 64, 32, 24, 64, 10), (20, 64, 64, 32, 16, 16, 50, 10),
                                                   (12, 64, 64, 32, 17, 16, 64, 10), (30, 200, 256, 128, 64, 64, 40, 12),
                                                   (12, 64, 64, 33, 16, 16, 64, 10)):
        args = sgpr_args()
        args.filters_1, args.filters_2, args.filters_3, args.tensor_neurons, args.bottle_neck_neurons = f1, f2, f3, t, bn
        args.node_num, args.K = node_num, K
        torch.manual_seed(labels * 1000 + f1 + t)
        model, sd = _randomised(sg_net.SG(args, labels))
        eng = model.engine()
        assert eng.any_shape and eng.pw == f3
        # (at least K padding slots per graph, like every graph of the reference's data: one-hot rows tie EXACTLY across
        # labels, and with fewer pads than K torch.topk's unspecified tie order would pick the neighbours)
        c, l = _label_sorted_graphs(24, node_num, node_num // 3, node_num - K, labels, seed=labels + f3)
        feats = torch.from_numpy(synth.dense_features(c, l, num_labels=labels))
        f_a, f_b = feats[:12], feats[12:]
        want, wa1, wa2 = oracle.forward(sd, f_a, f_b, K)
        got, a1, a2 = model({"features_1": f_a, "features_2": f_b})
        tag = (labels, f1, f2, f3, t, bn)
        assert got.shape == (12,) and a1.shape == (12, node_num, 1)
        assert (got.cpu() - want).abs().max().item() < SCORE_TOL, tag
        assert (a1.cpu() - wa1).abs().max().item() < 1e-4 and (a2.cpu() - wa2).abs().max().item() < 1e-4, tag
        emb = model.dgcnn_conv_pass(feats)
        ref_emb = oracle.conv_pass(sd, feats, K)
        assert emb.shape == (24, node_num, f3)
        assert (emb.cpu() - ref_emb).abs().max().item() < 1e-4 * max(1.0, float(ref_emb.abs().max())), tag
        # packed and ragged input: the same bits as the dense path's pooled vectors
        pooled, att, _ = model.embed(c, l, want_att=True)
        ref_pooled = oracle.embed(sd, feats, K)[0]
        assert pooled.shape == (24, f3)
        assert (pooled.cpu() - ref_pooled).abs().max().item() < 2e-4 * max(1.0, float(ref_pooled.abs().max())), tag
        rag = RaggedGraphs.from_padded(c, l, device="cuda", num_labels=labels)
        pooled_r, att_r, _ = model.embed(rag, None, want_att=True)
        assert torch.equal(pooled_r, pooled) and torch.equal(att_r, att)
        dense_pooled = eng.embed_dense(feats, K)[0]
        assert torch.equal(dense_pooled, pooled)
        # tails: all pairs, an index list long enough for the grouped kernel (which such a handle does not have), pairs
        mat = model.score_all_pairs(pooled, pooled)
        ref_mat = oracle.score_all_pairs(sd, ref_pooled, ref_pooled)
        assert (mat.cpu() - ref_mat).abs().max().item() < SCORE_TOL, tag
        rng = np.random.default_rng(labels)
        i1 = rng.integers(0, 24, 3000).astype(np.int32)
        i2 = rng.integers(0, 24, 3000).astype(np.int32)
        lst = model.score_pooled(pooled, pooled, torch.from_numpy(i1), torch.from_numpy(i2))
        # (the list kernel sums a pair's bilinear form in another order than the rectangle's row-hoisted form: fp32 rounding)
        assert (lst - mat[torch.from_numpy(i1).long().cuda(), torch.from_numpy(i2).long().cuda()]).abs().max().item() < 5e-5
        assert (lst.cpu() - ref_mat[torch.from_numpy(i1).long(), torch.from_numpy(i2).long()]).abs().max().item() < SCORE_TOL
        pairs = model.score_pooled(pooled[:12], pooled[12:])
        assert (pairs - mat[torch.arange(12), torch.arange(12, 24)]).abs().max().item() < 5e-5
        outs = eng.score_all_pairs_multi([(pooled[:7], pooled), (pooled[7:], pooled[:5])])
        assert torch.equal(outs[0], mat[:7]) and torch.equal(outs[1], mat[7:, :5])
        # the grouped entry point on an any-shape handle: the plan walked pair by pair - the bits of sgpr_score_pairs
        grouped = eng.score_pair_list(pooled, pooled, eng.pair_plan(i1, i2, 24, 24))
        assert torch.equal(grouped, eng.score_pairs(pooled, pooled, torch.from_numpy(i1), torch.from_numpy(i2))), tag
        with pytest.raises(SgprError, match="SGPR_E_DIMS"):
            eng.embed(c, l, K, debug=True)
        eng.check_status()
        bad = l.copy()
        bad[3, 0] = labels
        model.embed(torch.from_numpy(c).cuda(), torch.from_numpy(bad).cuda())
        with pytest.raises(SgprError):
            eng.check_status()
        # the stand-alone modules with this model's widths (layers_batch.py mirror: any width through the *_any entry points)
        from sg_pr_amd.layers_batch import AttentionModule, TenorNetworkModule
        att_mod = AttentionModule(args).cuda()
        att_mod.load_state_dict({"weight_matrix": sd["attention.weight_matrix"]})
        e = torch.randn(3, 300, f3).cuda()
        rep, sig = att_mod(e)
        w_rep, w_sig = oracle.attention({"attention.weight_matrix": sd["attention.weight_matrix"]}, e.cpu())
        assert rep.shape == (3, f3, 1) and (rep.cpu() - w_rep).abs().max().item() < 1e-4 * max(1.0, float(w_rep.abs().max()))
        assert (sig.cpu() - w_sig).abs().max().item() < 1e-5
        ntn_mod = TenorNetworkModule(args).cuda()
        ntn_mod.load_state_dict({k.split(".", 1)[1]: v for k, v in sd.items() if k.startswith("tensor_network.")})
        e1, e2 = torch.randn(7, f3, 1).cuda(), torch.randn(7, f3, 1).cuda()
        ntn = ntn_mod(e1, e2)
        w_ntn = oracle.tensor_network(sd, e1.cpu(), e2.cpu())
        assert ntn.shape == (7, t, 1) and (ntn.cpu() - w_ntn).abs().max().item() < 1e-4 * max(1.0, float(w_ntn.abs().max()))
    # dgcnn.knn beyond the LDS-resident kernel's 256 nodes / 32 neighbours: coordinate keys are the reference's bit for bit
    from sg_pr_amd import dgcnn
    gen = torch.Generator().manual_seed(3)
    for n, k, ch in ((700, 10, 3), (1024, 64, 3), (300, 40, 3), (100, 33, 3)):
        x = (torch.rand(2, ch, n, generator=gen) - 0.5) * 100.0
        got = dgcnn.knn(x.cuda(), k).cpu()
        pd = oracle.neg_sq_dist(x)
        want = oracle.knn(x, k)
        assert got.shape == (2, n, k) and got.dtype == torch.int64
        # same keys in the same order (indices may differ only between candidates whose reference keys are EQUAL)
        assert torch.equal(torch.gather(pd, 2, got), torch.gather(pd, 2, want)), (n, k)


def test_matrix_core_any_shape_embed(oracle):
    """sgpr_wide.hip: an any-shape handle inside its limits (labels <= 32, filters <= 128 / 128 / 64, node_num <= 112, K = 10)
    embeds on the matrix cores (two f16 planes per operand), everything else - and a graph whose values leave the f16
    range - on the plain-fp32 kernel (debug bit 23 forces that one).  Both against the oracle and against each other;
    packed, ragged and dense input give the same bits; the fallback's results ARE the plain kernel's."""
    from sg_pr_amd import sg_net, synth
    from sg_pr_amd.allpairs import RaggedGraphs
    from sg_pr_amd.parser_sg import sgpr_args
    for labels, f1, f2, f3, node_num in ((12, 128, 128, 64, 100), (25, 80, 112, 48, 112), (13, 64, 64, 32, 37)):
        args = sgpr_args()
        args.filters_1, args.filters_2, args.filters_3, args.tensor_neurons, args.bottle_neck_neurons = f1, f2, f3, 16, 16
        args.node_num, args.K = node_num, 10
        torch.manual_seed(labels + f1)
        model, sd = _randomised(sg_net.SG(args, labels))
        eng = model.engine()
        assert eng.any_shape and eng.pw == f3
        c, l = _label_sorted_graphs(40, node_num, node_num // 3, node_num - 10, labels, seed=labels + f3)
        feats = torch.from_numpy(synth.dense_features(c, l, num_labels=labels))
        ref_pooled, ref_att = oracle.embed(sd, feats, 10)[:2]
        ref_emb = oracle.conv_pass(sd, feats, 10)
        scale = max(1.0, float(ref_pooled.abs().max()))
        wide = eng.embed(c, l, 10, want_att=True, want_emb=True)
        eng.set_skip_mask(1 << 23)
        try:
            plain = eng.embed(c, l, 10, want_att=True, want_emb=True)
        finally:
            eng.set_skip_mask(0)
        tag = (labels, f1, f2, f3, node_num)
        for got in (wide, plain):
            assert (got[0].cpu() - ref_pooled).abs().max().item() < 2e-4 * scale, tag
            assert (got[1].cpu() - ref_att.reshape(40, node_num)).abs().max().item() < 1e-4, tag
            assert (got[2].cpu() - ref_emb).abs().max().item() < 1e-4 * max(1.0, float(ref_emb.abs().max())), tag
        assert not torch.equal(wide[0], plain[0])                      # (two datapaths, not one)
        assert (wide[0] - plain[0]).abs().max().item() < 1e-4 * scale, tag
        # the other input forms: the same bits
        rag = RaggedGraphs.from_padded(c, l, device="cuda", num_labels=labels)
        assert torch.equal(model.embed(rag, None)[0], wide[0])
        assert torch.equal(eng.embed_dense(feats, 10)[0], wide[0])
        # values outside the f16 range: the graph is flagged and embedded by the plain-fp32 kernel in the same call
        big = c.copy()
        big[3] *= 3000.0                                               # coordinates of ~1e5: beyond f16's 65504
        mixed = eng.embed(big, l, 10)[0]
        eng.set_skip_mask(1 << 23)
        try:
            plain_big = eng.embed(big, l, 10)[0]
        finally:
            eng.set_skip_mask(0)
        assert torch.equal(mixed[3], plain_big[3]) and torch.isfinite(mixed).all(), tag
        keep = [g for g in range(40) if g != 3]
        assert torch.equal(mixed[keep], wide[0][keep]), tag
        eng.check_status()
        # the dense all-pairs tail of such a handle: matrix cores (two f16 planes) against the plain-fp32 kernel and the oracle
        # on the SAME pooled vectors (at two scales: the randomised weights saturate the sigmoid at the first); ragged
        # sizes; inputs outside the f16 range leave the rectangle to the plain kernel (its bits)
        differs, spread = False, 0.0
        for sc_ in (1.0, 0.03):
            pooled = (wide[0] * sc_).contiguous()
            mat = eng.score_all_pairs(pooled, pooled)
            ref_mat = oracle.score_all_pairs(sd, pooled.cpu(), pooled.cpu())
            eng.set_skip_mask(1 << 23)
            try:
                mat_plain = eng.score_all_pairs(pooled, pooled)
            finally:
                eng.set_skip_mask(0)
            differs = differs or not torch.equal(mat, mat_plain)
            spread = max(spread, float(mat.max() - mat.min()))
            assert (mat - mat_plain).abs().max().item() < 2e-5, (tag, sc_)
            assert (mat.cpu() - ref_mat).abs().max().item() < SCORE_TOL, (tag, sc_)
            for r_, c_ in ((1, 1), (17, 33), (40, 16), (5, 40)):
                sub = eng.score_all_pairs(pooled[:r_].contiguous(), pooled[:c_].contiguous())
                assert torch.equal(sub, mat[:r_, :c_]), (tag, sc_, r_, c_)
            outs = eng.score_all_pairs_multi([(pooled[:7], pooled), (pooled[7:], pooled[:5])])
            assert torch.equal(outs[0], mat[:7]) and torch.equal(outs[1], mat[7:, :5]), (tag, sc_)
        assert differs and spread > 1e-3, (tag, spread)                # (two datapaths, and not only saturated scores)
        pooled = wide[0]
        eng.set_skip_mask(1 << 23)
        try:
            big_plain = eng.score_all_pairs(pooled[:7] * 3000.0, pooled * 3000.0)
        finally:
            eng.set_skip_mask(0)
        assert torch.equal(eng.score_all_pairs(pooled[:7] * 3000.0, pooled * 3000.0), big_plain), tag


def test_node_num_and_k_beyond_the_tuned_kernels(eng, oracle_sd, oracle):
    """parser_sg.py:19-22: node_num and K are free.  Beyond the tuned kernels' 256 slots / 32 neighbours the shipped
    checkpoint embeds on the any-shape kernel (pooled rows keep the handle's width, so the tuned tails score them);
    against the oracle, and against the tuned kernel where both serve a shape."""
    import sys
    from sg_pr_amd import synth
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from tie_proof import prove_graph
    for node_num, K, lo, hi in ((300, 10, 200, 290), (100, 40, 30, 60), (512, 48, 100, 464), (1024, 10, 900, 1000)):
        c, l = _label_sorted_graphs(6, node_num, lo, hi, 12, seed=node_num + K)
        feats = torch.from_numpy(synth.dense_features(c, l))
        pooled, att, emb = eng.embed(c, l, K, want_att=True, want_emb=True)
        ref_emb = oracle.conv_pass(oracle_sd, feats, K)
        ref_pooled, ref_att = oracle.embed(oracle_sd, feats, K)[:2]
        assert pooled.shape == (6, 32) and emb.shape == (6, node_num, 32)
        # a graph agrees to rounding, or the whole of its deviation is a proven near-tie of the reference's own fp32 keys
        # (a thousand nodes in 100 m x 100 m: |x|^2 ~ 2500 leaves the expanded distance ~5e-4 of rounding, the gap between
        # a row's 10th and 11th neighbour is a few m^2 - a handful of rows per graph tie; tests/tie_proof.py)
        dev = (emb.cpu() - ref_emb).abs().amax(dim=(1, 2))
        ok = dev < 2e-5             # (rounding level; a flip with a smaller effect than 1e-4 per node still moves a sum over 1024)
        for g in np.flatnonzero(~ok.numpy()):
            rep = prove_graph(eng, oracle, oracle_sd, c[g], l[g], K, pooled[g].cpu().numpy())
            assert rep["proven"], (node_num, K, int(g), rep["reason"])
        assert int(ok.sum()) >= (1 if node_num == 1024 else 3), (node_num, K, dev)
        assert (pooled.cpu() - ref_pooled)[ok].abs().max().item() < 2e-4 * max(1.0, float(ref_pooled.abs().max())), (node_num, K)
        assert (att.cpu() - ref_att.reshape(6, node_num))[ok].abs().max().item() < 1e-4
        got, a1, a2 = eng.forward_dense(feats[:3], feats[3:], K)
        assert a1.shape == (3, node_num) and torch.equal(torch.cat((a1, a2)), att)
        assert torch.equal(got, eng.score_pairs(pooled[:3].contiguous(), pooled[3:].contiguous()))
        mat = eng.score_all_pairs(pooled, pooled)
        ref_mat = oracle.score_all_pairs(oracle_sd, ref_pooled, ref_pooled)
        both = ok[:, None] & ok[None, :]
        # the tail on the same pooled vectors, then end to end.  (Hundreds of nodes pool into vectors of magnitude ~1e3 and
        # tensor-network pre-activations of ~1e5: the fp32 rounding of the pooled sums alone - the oracle's own summation
        # order is as arbitrary as ours - moves a score by more than 1e-4 there; the end-to-end gate is 1e-4 up to 300 slots,
        # beyond that the pooled vectors are held to 2e-4 relative above and the tail to 1e-4 on equal inputs)
        ours = pooled.cpu()
        assert (mat.cpu() - oracle.score_all_pairs(oracle_sd, ours, ours)).abs().max().item() < SCORE_TOL, (node_num, K)
        e2e = (mat.cpu() - ref_mat)[both].abs().max().item()
        assert both.any() and e2e < (SCORE_TOL if node_num <= 300 else 1e-3), (node_num, K, e2e)
    # where both kernels serve a shape they agree to rounding: K = 33 vs the tuned kernel is not possible, so compare the
    # any-shape model of the checkpoint on an any-shape handle's terms - a 13-label copy of the checkpoint (one extra,
    # unused label channel with zero weights) is an any-shape handle computing the same function
    from sg_pr_amd.engine import Engine, SgprDims
    sd13 = {k: v.clone() for k, v in oracle_sd.items()}
    w = sd13["dgcnn_f_conv1.0.weight"]
    w4 = w.reshape(w.shape[0], 2, 12)
    sd13["dgcnn_f_conv1.0.weight"] = torch.cat((w4, torch.zeros(w.shape[0], 2, 1)), dim=2).reshape(w.shape[0], 26, 1, 1)
    any13 = Engine(sd13, SgprDims(13, 64, 64, 32, 16, 16))
    assert any13.any_shape and any13.pw == 32
    c, l = _label_sorted_graphs(40, 100, 30, 95, 12, seed=77)
    p_any = any13.embed(c, l, 10)[0]
    p_tuned = eng.embed(c, l, 10)[0]
    assert (p_any - p_tuned).abs().max().item() < 2e-4 * max(1.0, float(p_tuned.abs().max()))
    m_any = any13.score_all_pairs(p_tuned, p_tuned)
    m_tuned = eng.score_all_pairs(p_tuned, p_tuned)
    assert (m_any - m_tuned).abs().max().item() < SCORE_TOL

