"""Host-side logic, C-ABI surface and the multi-process (gloo) sharding path.  CPU only."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------ C-ABI surface
def test_library_exports_every_header_symbol():
    from sg_pr_amd import engine
    lib = engine.load_library()
    header = open(os.path.join(REPO, "include", "sgpr.h")).read()
    declared = set(re.findall(r"\b(sgpr_[a-z0-9_]+)\s*\(", header))
    assert declared == set(engine.ABI_SYMBOLS), declared ^ set(engine.ABI_SYMBOLS)
    for sym in declared:
        assert getattr(lib, sym) is not None
    assert lib.sgpr_abi_version() == int(re.search(r"#define SGPR_ABI_VERSION (\d+)", header).group(1)) >= 2
    assert lib.sgpr_weights_count(ctypes.byref(engine.default_dims())) == 48689   # 48 696 minus 7 int64 counters


def test_create_rejects_bad_blobs_without_a_gpu():
    from sg_pr_amd import engine
    lib = engine.load_library()
    h = ctypes.c_void_p()
    blob = np.zeros(100, dtype=np.float32)
    rc = lib.sgpr_create(blob.ctypes.data_as(ctypes.c_void_p), blob.size, ctypes.byref(engine.default_dims()), 0,
                         ctypes.byref(h))
    assert rc == -8 and b"48689" in lib.sgpr_last_error()
    odd = engine.SgprDims(12, 64, 64, 64, 16, 16)          # larger than the built shape: an any-shape handle, same blob rule
    rc = lib.sgpr_create(blob.ctypes.data_as(ctypes.c_void_p), blob.size, ctypes.byref(odd), 0, ctypes.byref(h))
    assert rc == -8 and b"expected" in lib.sgpr_last_error()
    for beyond in (engine.SgprDims(65, 64, 64, 32, 16, 16), engine.SgprDims(12, 257, 64, 32, 16, 16),
                   engine.SgprDims(12, 64, 64, 129, 16, 16), engine.SgprDims(12, 64, 64, 32, 65, 16),
                   engine.SgprDims(12, 64, 64, 32, 16, 0)):
        rc = lib.sgpr_create(blob.ctypes.data_as(ctypes.c_void_p), blob.size, ctypes.byref(beyond), 0, ctypes.byref(h))
        assert rc == -2                                     # SGPR_E_DIMS: beyond the SGPR_ANY_MAX_* limits
    assert lib.sgpr_create(None, 0, ctypes.byref(odd), 0, ctypes.byref(h)) == -1


def test_lds_plans():
    from sg_pr_amd import engine
    lib = engine.load_library()
    zeroed = ctypes.create_string_buffer(1 << 16)   # plan queries read plain fields of the handle: a zeroed one is a
    h = ctypes.cast(zeroed, ctypes.c_void_p)        # handle of the built shape on a GPU of 0 CUs
    for n, k in [(64, 10), (100, 10), (256, 20), (16, 10), (128, 32), (200, 32), (23, 5)]:
        b = lib.sgpr_embed_lds_bytes(h, n, k)
        assert 0 < b <= 160 * 1024, (n, k, b)
    # > SGPR_MAX_NODES / > SGPR_MAX_K: the any-shape kernel - two rows of <= 128 floats + its working memory when that fits
    # LDS (the zeroed handle describes a model of width 0: squared norms + neighbour lists + 64 floats)
    assert lib.sgpr_embed_lds_bytes(h, 257, 10) == 1024 + (257 * 11 + 64) * 4
    assert lib.sgpr_embed_lds_bytes(h, 100, 33) == 1024 + (100 * 34 + 64) * 4
    assert lib.sgpr_embed_lds_bytes(h, 1025, 10) == 0     # > SGPR_ANY_MAX_NODES
    assert lib.sgpr_embed_lds_bytes(h, 100, 65) == 0      # > SGPR_ANY_MAX_K
    assert lib.sgpr_embed_lds_bytes(h, 8, 10) == 0        # K > node_num
    assert lib.sgpr_pooled_width(h) == 32 and lib.sgpr_is_any_shape(h) == 0
    assert lib.sgpr_pooled_width(None) == 0
    # one redo flag per launch slot + the second pass's request word (rounded to 256 B) + the parked first-branch block
    # for node_num > 128 + the split launch's region: one flag (the array rounded to 16 B) + 16 sem3 rows per launch slot,
    # for at most 128 slots (a split launch has at most half as many graphs as the GPU has CUs)
    sem = lambda g: (min(g, 128) * 8 + 15) // 16 * 16 + min(g, 128) * 16 * 32 * 4
    assert lib.sgpr_embed_workspace_bytes(h, 10, 100, 10) == 256 + sem(10)
    assert lib.sgpr_embed_workspace_bytes(h, 250, 100, 10) == 512 + sem(250)      # 250 flags + 8 bytes > 256
    assert lib.sgpr_embed_workspace_bytes(h, 300, 100, 10) == 512 + sem(300)
    assert lib.sgpr_embed_workspace_bytes(h, 100000, 100, 10) == (100008 + 255) // 256 * 256 + sem(128)
    assert lib.sgpr_embed_workspace_bytes(h, 10, 256, 20) == 256 + 10 * 256 * 32 * 4 + sem(10)


def test_engine_refuses_to_run_without_gpu(ckpt_path):
    from sg_pr_amd import engine
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    sd = torch.load(ckpt_path, map_location="cpu")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        engine.Engine(sd)


def test_blob_order_matches_checkpoint(ckpt_path):
    from sg_pr_amd import engine
    sd = torch.load(ckpt_path, map_location="cpu")
    blob = engine.blob_from_state_dict(sd)
    assert blob.dtype == np.float32 and blob.size == 48689
    w = sd["module.dgcnn_s_conv1.0.weight"].reshape(-1).numpy()
    np.testing.assert_array_equal(blob[: w.size], w)
    np.testing.assert_array_equal(blob[-1:], sd["module.scoring_layer.bias"].numpy())
    np.testing.assert_array_equal(blob[-17:-1], sd["module.scoring_layer.weight"].reshape(-1).numpy())


# ------------------------------------------------------------------ reference-shaped host API
def _write_config(tmp_path, **over):
    cfg = """
common:
  model: "%(model)s"
  cuda: "0"
  batch_size: 128
  p_thresh: 3
  graph_pairs_dir: "%(graphs)s"
  pair_list_dir: '%(lists)s'
arch:
  keep_node: 1
  filters_1: 64
  filters_2: 64
  filters_3: 32
  tensor_neurons: 16
  bottle_neck_neurons: 16
  K: 10
train:
  epochs: 500
  train_sequences: ['00']
  eval_sequences: ["08",]
  dropout: 0
  learning_rate: 0.001
  weight_decay: 0.0005
  gpu: 0
  logdir: "./logs_k10"
  node_num: 100
eva_batch:
  sequences: ["00",]
  output_path: "%(out)s"
  show: False
eva_pair:
  pair_file: ["%(graphs)s/0.json","%(graphs)s/250.json"]
""" % over
    path = tmp_path / "config.yml"
    path.write_text(cfg)
    return str(path)


def test_parser_reads_reference_layout(tmp_path, golden_dir, ckpt_path):
    from sg_pr_amd.parser_sg import sgpr_args
    cfg = _write_config(tmp_path, model=ckpt_path, graphs=os.path.join(golden_dir, "data"), lists=str(tmp_path),
                        out=str(tmp_path / "eva"))
    a = sgpr_args()
    assert a.node_num == 100 and a.K == 10 and a.batch_size == 128          # reference defaults
    a.load(cfg)
    assert a.cuda == "0" and a.gpu == 0 and a.p_thresh == 3 and a.sequences == ["00"]
    assert isinstance(a.pair_file, list) and len(a.pair_file) == 2
    with pytest.raises(KeyError):
        bad = tmp_path / "bad.yml"
        bad.write_text("common: {}\narch: {}\ntrain: {}\neva_batch: {}\neva_pair: {}\n")
        sgpr_args().load(str(bad))


def test_process_pair_and_pair_list(tmp_path, golden_dir):
    from sg_pr_amd.utils import process_pair, load_paires
    d = process_pair([os.path.join(golden_dir, "data", "0.json"), os.path.join(golden_dir, "data", "250.json")])
    assert set(d) == {"centers_1", "nodes_1", "centers_2", "nodes_2", "distance"}
    assert abs(d["distance"] - 133.12761323772054) < 1e-9 and len(d["nodes_1"]) == 38 and len(d["nodes_2"]) == 31
    lst = tmp_path / "00.txt"
    lst.write_text("0.json 3.json\n3.json 250.json\n")
    assert load_paires(str(lst), "/g") == [["/g/0.json", "/g/3.json"], ["/g/3.json", "/g/250.json"]]


def test_pack_graph_semantics():
    from sg_pr_amd.sg_net import pack_graph
    c, l = pack_graph([[1, 2, 3], [4, 5, 6]], [3, 11], 5)
    assert c.dtype == np.float32 and l.dtype == np.int32
    np.testing.assert_array_equal(l, [3, 11, -1, -1, -1])
    np.testing.assert_array_equal(c[2:], 0)
    with pytest.raises(KeyError):
        pack_graph([[0, 0, 0]], [12], 5)          # reference: KeyError at global_labels[node]
    with pytest.raises(ValueError):
        pack_graph(np.zeros((6, 3)), [0] * 6, 5, strict=True)
    # oversized graphs: the reference subsamples with the unseeded global RNG (sg_net.py:252-256); the engine's rule is
    # seeded by the graph itself -> same sorted subset every time, whatever else was packed before
    from sg_pr_amd.sg_net import subsample_indices
    rng = np.random.default_rng(5)
    cen, lab = rng.normal(size=(9, 3)), rng.integers(0, 12, size=9)
    with pytest.warns(UserWarning):
        c1, l1 = pack_graph(cen, lab, 5)
    pack_graph(rng.normal(size=(7, 3)), [1] * 7, 5)
    with pytest.warns(UserWarning):
        c2, l2 = pack_graph(cen.tolist(), lab.tolist(), 5)
    np.testing.assert_array_equal(c1, c2)
    np.testing.assert_array_equal(l1, l2)
    keep = subsample_indices(cen, lab, 5)
    assert len(keep) == 5 and (np.diff(keep) > 0).all()
    np.testing.assert_array_equal(l1, lab[keep])
    np.testing.assert_array_equal(c1, cen[keep].astype(np.float32))
    c, l = pack_graph([], [], 4)
    assert (l == -1).all()


def test_sg_module_state_dict_and_transfer(golden_dir, ckpt_path):
    from sg_pr_amd import sg_net
    from sg_pr_amd.parser_sg import sgpr_args
    from sg_pr_amd.utils import process_pair
    a = sgpr_args()
    a.model = ckpt_path
    t = sg_net.SGTrainer(a, False)
    raw = torch.load(ckpt_path, map_location="cpu")
    assert list(t.model.state_dict().keys()) == [k[7:] for k in raw.keys()]     # strict layout, same order
    for k, v in raw.items():
        assert torch.equal(t.model.state_dict()[k[7:]], v)
    assert t.model.module is t.model and not t.model.training
    g = np.load(os.path.join(golden_dir, "kitti3_n100_k10.npz"))
    d = process_pair([os.path.join(golden_dir, "data", "0.json"), os.path.join(golden_dir, "data", "250.json")])
    x = t.transfer_to_torch(d, False)
    assert x["features_1"].dtype == np.float64 and x["features_1"].shape == (15, 100) and x["target"] == 0.0
    np.testing.assert_array_equal(x["features_1"].astype(np.float32), g["features"][0])
    np.testing.assert_array_equal(x["features_2"].astype(np.float32), g["features"][2])
    assert len(d["nodes_1"]) == 38                                             # input not mutated (reference mutates)
    with pytest.raises(SystemExit):
        t.target_from_distance(10.0)                                            # 3 m < d < 20 m: reference exit(-1)
    with pytest.raises(NotImplementedError):
        sg_net.SGTrainer(a, True)


def test_metrics_golden(golden_dir):
    from sg_pr_amd import metrics
    g = np.load(os.path.join(golden_dir, "prf1.npz"))
    for c in range(int(g["ncases"])):
        p, r, _ = metrics.precision_recall_curve(g[f"gt{c}"], g[f"score{c}"])
        np.testing.assert_allclose(p, g[f"precision{c}"], atol=1e-12)
        np.testing.assert_allclose(r, g[f"recall{c}"], atol=1e-12)
        assert abs(metrics.f1_max(g[f"gt{c}"], g[f"score{c}"]) - float(g[f"f1max{c}"])) < 1e-12
        assert abs(metrics.roc_auc(g[f"gt{c}"], g[f"score{c}"]) - float(g[f"auc{c}"])) < 1e-9


def test_roc_curve_golden(golden_dir):
    """eval_batch.py:48-49 - roc_curve (collinear points dropped, +inf first threshold) and auc vs sklearn's."""
    from sg_pr_amd import metrics
    g = np.load(os.path.join(golden_dir, "prf1.npz"))
    for c in range(int(g["ncases"])):
        fpr, tpr, thr = metrics.roc_curve(g["gt%d" % c], g["score%d" % c])
        np.testing.assert_allclose(fpr, g["fpr%d" % c], rtol=0, atol=1e-15)
        np.testing.assert_allclose(tpr, g["tpr%d" % c], rtol=0, atol=1e-15)
        np.testing.assert_array_equal(thr, g["roc_thr%d" % c].astype(thr.dtype))
        assert abs(metrics.auc(fpr, tpr) - float(g["auc%d" % c])) < 1e-12
        assert abs(metrics.roc_auc(g["gt%d" % c], g["score%d" % c]) - float(g["auc%d" % c])) < 1e-12


def test_roc_auc_from_counts(golden_dir):
    """The sort-free ROC area (every negative ranked among the distinct positive scores) equals sklearn's, ties included."""
    from sg_pr_amd import metrics
    g = np.load(os.path.join(golden_dir, "prf1.npz"))
    for c in range(int(g["ncases"])):
        for budget in (metrics.MAX_THRESHOLDS, 5):
            _, auc, _ = metrics.pr_roc_from_counts(*metrics.counts_of(g["score%d" % c], g["gt%d" % c]), max_thresholds=budget)
            assert abs(auc - float(g["auc%d" % c])) < 1e-12, c
    # saturated scores (sigmoid outputs pile up next to 0 and 1) and ignored pairs
    rng = np.random.default_rng(4)
    gt = rng.integers(-1, 2, size=20000)
    sc = (1 / (1 + np.exp(-rng.normal(3 * (gt == 1), 4)))).astype(np.float32)
    sc[rng.random(20000) < 0.2] = 1.0
    keep = gt >= 0
    for budget in (metrics.MAX_THRESHOLDS, 64):
        _, auc, _ = metrics.pr_roc_from_counts(*metrics.counts_of(sc, gt), max_thresholds=budget)
        assert abs(auc - metrics.roc_auc(gt[keep], sc[keep])) < 1e-12


def test_every_shipped_checkpoint_loads_strictly(release_state_dicts, ckpt_path, oracle):
    """`load_state_dict` of any of the 19 shipped checkpoints must succeed strictly, and each flattens
    into the C-ABI blob."""
    from sg_pr_amd import sg_net, engine
    from sg_pr_amd.parser_sg import sgpr_args
    model = sg_net.SG(sgpr_args(), 12)
    sds = dict(release_state_dicts)
    sds["model.pth"] = oracle.load_checkpoint(ckpt_path)
    assert len(sds) == 19
    blobs = []
    for name, sd in sds.items():
        model.load_state_dict(sd, strict=True)
        blob = engine.blob_from_state_dict(model.state_dict())
        assert blob.shape == (48689,) and np.isfinite(blob).all(), name
        blobs.append(blob)
    assert len({b.tobytes() for b in blobs}) == 19          # nineteen different models


def test_synth_generators():
    from sg_pr_amd import synth
    c, l, n = synth.config2_pairs(seed=0)
    assert c.shape == (256, 64, 3) and l.shape == (256, 64) and n.max() <= 54 and n.min() >= 20
    c5, l5, n5 = synth.config5_pairs(seed=0, batch=4)
    assert c5.shape == (8, 256, 3) and (256 - n5 >= 20).all()
    ck, lk, nk, poses = synth.kitti_like_sequence(64, 100, 1)
    assert poses.shape == (64, 12) and ((lk >= 0).sum(1) == nk).all()
    d = synth.dense_features(c[:2], l[:2])
    assert d.shape == (2, 15, 64) and set(np.unique(d[:, 3:].sum(1))) <= {0.0, 1.0}
    c2, _, _ = synth.config2_pairs(seed=0)
    np.testing.assert_array_equal(c, c2)


# ------------------------------------------------------------------ all-pairs sharding (gloo, world_size 2)
def test_shard_bounds():
    from sg_pr_amd.allpairs import shard_bounds
    for total in (0, 1, 7, 4541):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def _allpairs_worker(rank, world, port, golden, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, REPO)
    from oracle import sgpr_oracle as oracle
    from sg_pr_amd import synth, allpairs
    torch.set_num_threads(2)
    sd = oracle.load_checkpoint(os.path.join(golden, "model.pth"))
    centers, labels, _, _ = synth.kitti_like_sequence(11, 100, 3)     # 11 graphs: uneven shards

    def embed_fn(c, l):
        if c.shape[0] == 0:
            return torch.empty(0, 32)
        return oracle.embed(sd, torch.from_numpy(synth.dense_features(c, l)), 10)[0]

    def padded_of(rag):                                      # a ragged shard back as the padded arrays embed_fn takes
        n = rag.node_num
        c = np.zeros((len(rag), n, 3), np.float32)
        l = -np.ones((len(rag), n), np.int32)
        off = (rag.offsets - rag.offsets[0]).numpy()
        for g in range(len(rag)):
            c[g, :off[g + 1] - off[g]] = rag.centers[off[g]:off[g + 1]].numpy()
            l[g, :off[g + 1] - off[g]] = rag.labels[off[g]:off[g + 1]].numpy()
        return c, l

    def embed_any(c, l):
        return embed_fn(*padded_of(c)) if isinstance(c, allpairs.RaggedGraphs) else embed_fn(c, l)

    scorer = allpairs.AllPairsScorer(embed_fn=embed_any, score_fn=lambda r, c: oracle.score_all_pairs(sd, r, c))
    # the same job from the ragged store (sgpr_embed_ragged's format): sharded by graph ranges like the padded arrays
    rag = allpairs.RaggedGraphs.from_padded(centers, labels)
    assert len(rag) == 11 and rag.shape == (11, 100) and len(rag[3:7]) == 4 and len(rag[9:]) == 2
    rag_full = scorer.run(rag, None, chunks=1)
    rag_set = allpairs.SequenceSet(scorer, [(rag, None), (rag[:5], None)]).run(chunks=1)
    full = scorer.run(centers, labels)                       # chunks = 4: pieces shipped while the next one is scored
    plain = scorer.run(centers, labels, chunks=1)            # the plain form: one block, one gather
    if rank == 0:
        torch.testing.assert_close(full, plain, rtol=0, atol=2e-6)   # torch-CPU matmuls are not batch-invariant
        torch.testing.assert_close(rag_full, plain, rtol=0, atol=2e-6)
        torch.testing.assert_close(rag_set[0], plain, rtol=0, atol=2e-6)
        assert rag_set[1].shape == (5, 5)
        torch.save(full, os.path.join(out_dir, "w%d.pt" % world))
    # several sequences as one job (allpairs.SequenceSet): the shards of both embedded by one call, matrices as before
    tc, tl = torch.from_numpy(centers), torch.from_numpy(labels)
    sset = allpairs.SequenceSet(scorer, [(tc, tl), (tc[:5], tl[:5])])
    many = sset.run(embed_fn=lambda c, l: embed_fn(c.numpy(), l.numpy()), chunks=1)
    short = scorer.run(centers[:5], labels[:5], chunks=1)
    if rank == 0:
        torch.testing.assert_close(many[0], plain, rtol=0, atol=2e-6)
        torch.testing.assert_close(many[1], short, rtol=0, atol=2e-6)
        assert many[1].shape == (5, 5)
    # sharded F1-max / ROC area without gathering: positives all-gathered, per-rank counts of the negatives (numpy
    # stand-in for the HIP pass) all-reduced
    from sg_pr_amd import metrics
    _, _, _, poses = synth.kitti_like_sequence(11, 100, 3)
    poses[:, [3, 11]] *= 4.0                                 # 4 m per frame: pairs beyond 20 m exist among 11 frames
    block = scorer.score_rows(scorer.pooled_all(centers, labels))

    def fns(blk, row0, xz):
        d = torch.cdist(xz[row0:row0 + blk.shape[0]].double(), xz.double())
        gt = torch.where(d <= 3, 1, torch.where(d >= 20, 0, -1)).numpy()
        return metrics.counts_of(blk.numpy(), gt)

    f1, auc = scorer.pr_roc(block, poses, fns=fns)
    assert f1 == scorer.f1_max(block, poses, fns=fns)
    f1 = "%r %r" % (f1, auc)
    with open(os.path.join(out_dir, "f1_w%d_r%d.txt" % (world, rank)), "w") as f:
        f.write(f1)
    # a failure in ONE rank's block (the engine raises ValueError for negative / NaN scores) reaches every rank before
    # any collective of the metric: the others raise too instead of blocking in the all_gather
    def bad_fns(blk, row0, xz):
        if rank == world - 1:
            raise ValueError("3 scores are negative or NaN")
        return fns(blk, row0, xz)

    with pytest.raises(ValueError if rank == world - 1 else RuntimeError, match="negative or NaN|another rank"):
        scorer.pr_roc(block, poses, fns=bad_fns)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_allpairs_two_ranks_equals_one(tmp_path, golden_dir):
    """The gathered matrix of a 2-rank run is bit-identical to the single-process one."""
    import torch.multiprocessing as mp
    from oracle import sgpr_oracle as oracle
    from sg_pr_amd import synth, allpairs
    mp.spawn(_allpairs_worker, args=(2, 29611, golden_dir, str(tmp_path)), nprocs=2, join=True)
    mp.spawn(_allpairs_worker, args=(3, 29613, golden_dir, str(tmp_path)), nprocs=3, join=True)   # uneven: 4 + 4 + 3 rows
    two = torch.load(os.path.join(str(tmp_path), "w2.pt"))
    three = torch.load(os.path.join(str(tmp_path), "w3.pt"))
    sd = oracle.load_checkpoint(os.path.join(golden_dir, "model.pth"))
    centers, labels, _, poses = synth.kitti_like_sequence(11, 100, 3)
    poses[:, [3, 11]] *= 4.0                                 # (as in the worker)
    torch.set_num_threads(2)
    # same shard shapes as the two ranks (torch-CPU matmuls are not bitwise batch-invariant; the HIP engine is,
    # which tests/test_gpu_parity.py checks on the GPU)
    def emb(lo, hi):
        return oracle.embed(sd, torch.from_numpy(synth.dense_features(centers[lo:hi], labels[lo:hi])), 10)[0]

    bounds = [allpairs.shard_bounds(11, 2, r) for r in range(2)]
    pooled = torch.cat([emb(lo, hi) for lo, hi in bounds])
    one = torch.cat([oracle.score_all_pairs(sd, pooled[lo:hi].contiguous(), pooled) for lo, hi in bounds])
    assert two.shape == (11, 11)
    assert torch.equal(one, two)
    # the oracle's torch-CPU embed is not bitwise batch-invariant (see above): 3 ranks embed other shard shapes
    torch.testing.assert_close(three, two, rtol=0, atol=2e-6)
    assert not torch.equal(two, two.t())                 # the NTN is asymmetric: full square needed
    gt, valid = allpairs.ground_truth_mask(allpairs.pose_distance_matrix(poses), 3)
    assert gt.shape == (11, 11) and valid.diagonal().all() and gt.diagonal().all() and (gt[valid] == 0).any()
    # both ranks computed the same F1-max / ROC area from their own row blocks, equal to the gathered-matrix values
    from sg_pr_amd import metrics
    want = metrics.f1_max(gt[valid].numpy(), two[valid].numpy())
    want_auc = metrics.roc_auc(gt[valid].numpy(), two[valid].numpy())
    for r in range(2):
        f1, auc = (float(v) for v in open(os.path.join(str(tmp_path), "f1_w2_r%d.txt" % r)).read().split())
        assert abs(f1 - want) < 1e-12
        assert abs(auc - want_auc) < 1e-12


class _OracleModel:
    """CPU stand-in for the two HIP entry points eval_batch uses (tests of the host / sharding logic only)."""

    def __init__(self, oracle, sd):
        self.oracle, self.sd = oracle, sd

    def embed(self, c, l):
        from sg_pr_amd import synth
        return self.oracle.embed(self.sd, torch.from_numpy(synth.dense_features(np.asarray(c), np.asarray(l))), 10)

    def score_pooled(self, p1, p2, i1, i2, grouped=None):
        return self.oracle.score_from_pooled(self.sd, p1[i1.long()], p2[i2.long()])

    def engine(self):
        return self

    def check_status(self):
        pass


def _pair_list_worker(rank, world, port, golden, out_dir, cfg, break_rank):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, REPO)
    from oracle import sgpr_oracle as oracle
    from sg_pr_amd import eval_batch, sg_net
    from sg_pr_amd.parser_sg import sgpr_args
    torch.set_num_threads(2)
    args = sgpr_args()
    args.load(cfg)
    trainer = sg_net.SGTrainer(args, False)
    trainer.model = _OracleModel(oracle, oracle.load_checkpoint(os.path.join(golden, "model.pth")))
    os.makedirs(args.output_path, exist_ok=True)
    if break_rank is None:
        f1 = eval_batch.evaluate_sequence(trainer, "00", args, plots=False)
        pairs = eval_batch.load_paires(os.path.join(args.pair_list_dir, "00.txt"), args.graph_pairs_dir)
        pred, gt = eval_batch.score_pair_list(trainer, pairs)
        with open(os.path.join(out_dir, "embedded_w%d_r%d.txt" % (world, rank)), "w") as f:
            f.write(str(eval_batch.score_pair_list.last_embedded))
        np.save(os.path.join(out_dir, "pred_w%d_r%d.npy" % (world, rank)), pred)
        np.save(os.path.join(out_dir, "gt_w%d_r%d.npy" % (world, rank)), gt)
        assert f1 == 1.0
    else:
        # a failure on ONE rank (a graph file that does not exist) is raised on EVERY rank, not a hang of the others
        pairs = eval_batch.load_paires(os.path.join(args.pair_list_dir, "00.txt"), args.graph_pairs_dir)
        lo, hi = __import__("sg_pr_amd.allpairs", fromlist=["x"]).shard_bounds(len(pairs), world, break_rank)
        pairs[lo][0] = pairs[lo][0] + ".missing"
        try:
            eval_batch.score_pair_list(trainer, pairs)
            outcome = "no error"
        except FileNotFoundError:
            outcome = "own"
        except RuntimeError as e:
            outcome = "other" if "another rank" in str(e) else "unexpected %r" % e
        with open(os.path.join(out_dir, "err_r%d.txt" % rank), "w") as f:
            f.write(outcome)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_pair_list_sharded_equals_single_process(tmp_path, golden_dir, ckpt_path, oracle, oracle_sd):
    """SURVEY 8e pair-list mode, graph-sharded: the distinct graphs of the list are split over the ranks (each rank parses
    and embeds G / world of them), one all-gather of the pooled vectors and of the poses, a contiguous split of the list for
    the tail, gather of float32[P_r] (gloo, world sizes 2 and 3 - 7 pairs over 3 graphs: uneven shards).  Every rank
    ends up with the single-process vectors; rank 0 alone writes the artefacts; a rank-local failure is raised by all
    ranks."""
    import torch.multiprocessing as mp
    from sg_pr_amd import eval_batch, sg_net
    from sg_pr_amd.parser_sg import sgpr_args
    data = os.path.join(golden_dir, "data")
    lines = ["0.json 250.json", "0.json 3.json", "250.json 250.json", "3.json 0.json", "250.json 0.json", "3.json 3.json",
             "0.json 0.json"]
    (tmp_path / "00.txt").write_text("\n".join(lines) + "\n")
    cfg = _write_config(tmp_path, model=ckpt_path, graphs=data, lists=str(tmp_path), out=str(tmp_path / "eva"))
    args = sgpr_args()
    args.load(cfg)
    trainer = sg_net.SGTrainer(args, False)
    trainer.model = _OracleModel(oracle, oracle_sd)
    pairs = eval_batch.load_paires(str(tmp_path / "00.txt"), data)
    want_pred, want_gt = eval_batch.score_pair_list(trainer, pairs)
    assert want_pred.dtype == np.float32 and want_gt.dtype == np.float64 and len(want_pred) == 7
    for world, port in ((2, 29621), (3, 29623)):
        mp.spawn(_pair_list_worker, args=(world, port, golden_dir, str(tmp_path), cfg, None), nprocs=world, join=True)
        from sg_pr_amd import allpairs
        embedded = [int(open(str(tmp_path / ("embedded_w%d_r%d.txt" % (world, r)))).read()) for r in range(world)]
        # every rank embedded ITS shard of the 3 distinct graphs (a contiguous split of the list alone would have had
        # each of the 2 ranks embed all 3)
        assert embedded == [b - a for a, b in (allpairs.shard_bounds(3, world, r) for r in range(world))], embedded
        for r in range(world):
            pred = np.load(str(tmp_path / ("pred_w%d_r%d.npy" % (world, r))))
            gt = np.load(str(tmp_path / ("gt_w%d_r%d.npy" % (world, r))))
            assert pred.dtype == np.float32 and gt.dtype == np.float64
            np.testing.assert_array_equal(gt, want_gt)
            # (torch-CPU matmuls are not bitwise batch-invariant: the oracle stand-in embeds other batch shapes per
            # rank; the HIP engine is batch-invariant and the GPU suite holds the sharded run to bit equality)
            np.testing.assert_allclose(pred, want_pred, rtol=0, atol=2e-6)
    np.testing.assert_allclose(np.load(str(tmp_path / "eva" / "00_DL_db.npy")), want_pred, rtol=0, atol=2e-6)
    assert float(open(str(tmp_path / "eva" / "00_DL_F1_max.txt")).read()) == 1.0
    mp.spawn(_pair_list_worker, args=(2, 29625, golden_dir, str(tmp_path), cfg, 1), nprocs=2, join=True)
    assert open(str(tmp_path / "err_r1.txt")).read() == "own"
    assert open(str(tmp_path / "err_r0.txt")).read() == "other"


def test_eval_batch_pair_list_logic(tmp_path, golden_dir, ckpt_path, oracle, oracle_sd):
    """eval_batch counterpart: graphs embedded once, index lists drive the tail; artefacts like the reference's."""
    from sg_pr_amd import eval_batch, sg_net, synth
    from sg_pr_amd.parser_sg import sgpr_args
    data = os.path.join(golden_dir, "data")
    (tmp_path / "00.txt").write_text("0.json 250.json\n0.json 3.json\n250.json 250.json\n3.json 0.json\n")
    cfg = _write_config(tmp_path, model=ckpt_path, graphs=data, lists=str(tmp_path), out=str(tmp_path / "eva"))
    args = sgpr_args()
    args.load(cfg)
    trainer = sg_net.SGTrainer(args, False)

    trainer.model = _OracleModel(oracle, oracle_sd)
    os.makedirs(args.output_path, exist_ok=True)
    f1 = eval_batch.evaluate_sequence(trainer, "00", args, plots=True)
    for png in ("00_DL_roc_curve.png", "00_DL_pr_curve.png"):                    # eval_batch.py:66, 80
        assert os.path.getsize(os.path.join(args.output_path, png)) > 1000
    pred = np.load(os.path.join(args.output_path, "00_DL_db.npy"))
    gt = np.load(os.path.join(args.output_path, "00_gt_db.npy"))
    assert pred.dtype == np.float32 and gt.dtype == np.float64
    g = np.load(os.path.join(golden_dir, "kitti3_n100_k10.npz"))
    want = [g["scores"][2], g["scores"][1], g["scores"][8], g["scores"][3]]
    np.testing.assert_allclose(pred, want, atol=1e-6)
    np.testing.assert_array_equal(gt, [0, 1, 1, 1])
    assert f1 == 1.0 and float(open(os.path.join(args.output_path, "00_DL_F1_max.txt")).read()) == 1.0


def test_f1_max_from_counts_is_exact(golden_dir):
    """Host half of the device-side F1-max: thresholds at the positives' scores, bounds in between, refinement of the
    segments that can still hold the maximum == sklearn-style sort, incl. ties and ignored pairs - whatever the
    threshold budget per pass is."""
    from sg_pr_amd import metrics
    g = np.load(os.path.join(golden_dir, "prf1.npz"))
    for c in range(int(g["ncases"])):
        for budget in (metrics.MAX_THRESHOLDS, 7, 1):
            got, _, _ = metrics.pr_roc_from_counts(*metrics.counts_of(g[f"score{c}"], g[f"gt{c}"]), max_thresholds=budget)
            assert abs(got - float(g[f"f1max{c}"])) < 1e-12
    rng = np.random.default_rng(7)
    for trial in range(6):
        n = 60000
        gt = (rng.random(n) < 0.03).astype(np.int64)
        sc = (1.0 / (1.0 + np.exp(-(rng.normal(0, 3, n) + 4 * gt)))).astype(np.float32)
        if trial == 1:
            sc = np.round(sc, 2).astype(np.float32)            # massive ties
        if trial == 2:
            sc[:] = 0.25                                        # one distinct value
        if trial == 3:
            gt[:] = 0                                           # no positives
        if trial == 4:
            sc = np.where(rng.random(n) < 0.5, np.float32(1.0), sc).astype(np.float32)   # saturated scores
        if trial == 5:
            gt = (rng.random(n) < 0.5).astype(np.int64)         # as many positives as negatives
        ign = rng.random(n) < 0.2
        for budget, max_passes in ((metrics.MAX_THRESHOLDS, 2), (100, 6), (16, 16)):
            got, auc, passes = metrics.pr_roc_from_counts(*metrics.counts_of(sc, np.where(ign, -1, gt)), max_thresholds=budget)
            assert abs(got - metrics.f1_max(gt[~ign], sc[~ign])) < 1e-12 and passes <= max_passes, (trial, budget, passes)
            if trial != 3:
                assert abs(auc - metrics.roc_auc(gt[~ign], sc[~ign])) < 1e-12


def test_custom_ops_registered_and_refuse_cpu():
    """SURVEY §8b custom ops: registered with the dispatcher, shape-traceable, and GPU-only."""
    from sg_pr_amd import ops  # noqa: F401
    from torch._subclasses.fake_tensor import FakeTensorMode
    for name in ("embed", "score_pairs", "score_all_pairs", "forward_dense"):
        assert hasattr(torch.ops.sgpr, name)
    with FakeTensorMode():
        c, l, w = torch.empty(5, 100, 3), torch.empty(5, 100, dtype=torch.int32), torch.empty(48689)
        p, a = torch.ops.sgpr.embed(c, l, w, 10)
        assert p.shape == (5, 32) and a.shape == (5, 100)
        assert torch.ops.sgpr.score_all_pairs(p, p, w).shape == (5, 5)
        assert torch.ops.sgpr.score_pairs(p, p, w).shape == (5,)
        f = torch.empty(4, 15, 64)
        s_, a1, a2 = torch.ops.sgpr.forward_dense(f, f, w, 10)
        assert s_.shape == (4,) and a1.shape == (4, 64) and a2.shape == (4, 64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        torch.ops.sgpr.score_pairs(torch.zeros(2, 32), torch.zeros(2, 32), torch.zeros(48689))


def test_graph_store_roundtrip(tmp_path, golden_dir):
    """a sequence directory is parsed once into the packed wire format (same bytes as pack_graph per
    graph, frames in natural order) and survives the .npz round trip."""
    from sg_pr_amd import graph_store, sg_net, utils
    data = os.path.join(golden_dir, "data")
    seq = graph_store.pack_directory(data, 100)
    assert seq.names == ["0.json", "3.json", "250.json"] and len(seq) == 3 and seq.node_num == 100
    for g, name in enumerate(seq.names):
        d = utils.read_graph(os.path.join(data, name))
        c, l = sg_net.pack_graph(d["centers"], d["nodes"], 100)
        np.testing.assert_array_equal(seq.centers[g], c)
        np.testing.assert_array_equal(seq.labels[g], l)
        np.testing.assert_array_equal(seq.poses[g], np.asarray(d["pose"], dtype=np.float64))
    seq.save(str(tmp_path / "s.npz"))
    back = graph_store.PackedSequence.load(str(tmp_path / "s.npz"))
    assert back.names == seq.names
    for a, b in ((seq.centers, back.centers), (seq.labels, back.labels), (seq.poses, back.poses)):
        np.testing.assert_array_equal(a, b)
    g = np.load(os.path.join(golden_dir, "kitti3_n100_k10.npz"))
    if "centers" in g and g["centers"].shape[0] == 3:
        pass


def test_processed_slots_and_size_order_logic():
    """Engine.processed_slots (torch) == synth.effective_nodes (numpy) - the promise behind node_cap / launch order."""
    from sg_pr_amd import engine, synth
    for n, k, lo, hi in ((100, 10, 25, 60), (64, 10, 20, 60), (40, 5, 1, 39), (256, 20, 100, 250)):
        c, l, n_real = synth.make_graphs(50, n, lo, hi, 3 + n)
        eff = engine.Engine.processed_slots(c, l, k).numpy()
        np.testing.assert_array_equal(eff, synth.effective_nodes(c, l, k))
        m = n - n_real
        np.testing.assert_array_equal(eff, n_real + np.where((m >= k) & (m > 1), 1, m))
        assert engine.Engine.node_cap_of(c, l, k) == int(eff.max())
    # a graph whose trailing duplicates are REAL nodes (same centre and label) is compressed the same way
    c = np.zeros((1, 12, 3), dtype=np.float32)
    l = np.zeros((1, 12), dtype=np.int32)
    c[0, :4] = np.arange(12, dtype=np.float32).reshape(4, 3)
    assert int(engine.Engine.processed_slots(c, l, 5)[0]) == 4 + 1        # 8 identical trailing slots >= k -> one kept
    assert int(engine.Engine.processed_slots(c, l, 9)[0]) == 12           # fewer than k copies: all kept


def _build_c_demo(tmp_path):
    import subprocess
    from sg_pr_amd import _build
    _build.build_library()
    exe = str(tmp_path / "sgpr_demo")
    cmd = ["gcc", "-O2", "-std=c11", "-D__HIP_PLATFORM_AMD__", os.path.join(REPO, "examples", "sgpr_demo.c"),
           "-I" + os.path.join(REPO, "include"), "-I/opt/rocm/include", "-L" + _build.LIB_DIR, "-lsgpr_hip",
           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + _build.LIB_DIR, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.run(cmd, check=True)
    return exe


def test_build_staleness_is_by_content_and_abi_is_checked(tmp_path, monkeypatch):
    """The library is reused exactly while the hash recorded next to it equals the hash of the sources on disk
    (no timestamps, no sniffing of the machine), and a library that answers another C-ABI version than include/sgpr.h
    is refused at load time."""
    import shutil
    from sg_pr_amd import _build, engine
    _build.build_library()
    assert not _build.is_stale()
    assert open(_build.LIB_PATH + ".srchash").read().strip() == _build.source_hash()
    real = _build.HEADERS[0]
    fake = tmp_path / "sgpr.h"
    fake.write_text(open(real).read().replace("#define SGPR_ABI_VERSION %d" % _build.header_abi_version(),
                                              "#define SGPR_ABI_VERSION 99"))
    monkeypatch.setattr(_build, "HEADERS", [str(fake)] + _build.HEADERS[1:])
    assert _build.is_stale() and _build.header_abi_version() == 99          # other header bytes -> other hash
    monkeypatch.setattr(engine, "_lib", None)
    with pytest.raises(ImportError, match="C-ABI version"):
        engine.load_library()
    monkeypatch.undo()
    assert not _build.is_stale()
    # newer timestamps alone change nothing
    os.utime(os.path.join(_build.CSRC, "sgpr_api.hip"))
    assert not _build.is_stale()


def test_c_abi_links_from_plain_c(tmp_path):
    """The boundary is a C ABI: a C11 translation unit including only sgpr.h + the HIP runtime API compiles with gcc
    and links against libsgpr_hip.so (examples/sgpr_demo.c; it is RUN by the GPU suite)."""
    exe = _build_c_demo(tmp_path)
    assert os.path.exists(exe)


def test_ragged_store_conversion_and_order():
    """Engine.to_ragged / ragged_order are host logic: counts, offsets, the processed-slot rule of sgpr_embed_capped."""
    from sg_pr_amd.engine import Engine
    from sg_pr_amd import synth
    c, l, _, _ = synth.kitti_like_sequence(50, 100, seed=1)
    rc, rl, off = Engine.to_ragged(c, l)
    counts = (l >= 0).sum(1)
    assert off[0] == 0 and (np.diff(off) == counts).all() and rc.shape == (counts.sum(), 3) and rl.dtype == np.int8
    for g in (0, 17, 49):
        assert (rc[off[g]:off[g + 1]] == c[g, :counts[g]]).all() and (rl[off[g]:off[g + 1]] == l[g, :counts[g]]).all()
    eff = Engine.processed_slots(c, l, 10).numpy()
    m = 100 - counts
    assert (eff == counts + np.where((m >= 10) & (m > 1), 1, m)).all()
    bad = l.copy()
    bad[3, 0] = -1                                              # a hole before real nodes
    with pytest.raises(ValueError):
        Engine.to_ragged(c, bad)
    # a label outside 0..11 is refused before the int8 cast could wrap it into a valid class (261 -> 5); the reference
    # raises KeyError (sg_net.py:277), the padded path flags it on the device
    for wrong in (12, 128, 261):
        bad = l.copy()
        bad[7, 2] = wrong
        with pytest.raises(ValueError, match="label"):
            Engine.to_ragged(c, bad)


def test_ragged_graphs_container():
    """allpairs.RaggedGraphs: slicing by graph range, concatenation, round trip through Engine.to_ragged."""
    from sg_pr_amd import allpairs, synth
    c, l, _, _ = synth.kitti_like_sequence(20, 100, seed=2)
    rag = allpairs.RaggedGraphs.from_padded(c, l)
    counts = (l >= 0).sum(1)
    assert len(rag) == 20 and int(rag.offsets[-1]) == counts.sum()
    part = rag[4:9]
    assert len(part) == 5 and int(part.offsets[-1] - part.offsets[0]) == counts[4:9].sum()
    assert torch.equal(part.centers, rag.centers[int(rag.offsets[4]):int(rag.offsets[9])])
    again = allpairs.RaggedGraphs.cat([rag[:4], part, rag[9:]])
    assert torch.equal(again.centers, rag.centers) and torch.equal(again.labels, rag.labels)
    assert torch.equal(again.offsets, rag.offsets)
    assert len(rag[7:7]) == 0 and len(allpairs.RaggedGraphs.cat([rag[7:7], rag[:2]])) == 2
    with pytest.raises(TypeError):
        rag[::2]
    with pytest.raises(ValueError):
        allpairs.RaggedGraphs(rag.centers, rag.labels, rag.offsets[:-1], 100)


def test_pair_plan_groups_a_list_by_row_graph():
    """sgpr_pair_plan (host half of sgpr_score_pair_list; no GPU work): distinct row graphs ascending, work items of <= 16
    pairs that never cross a row graph, a stable order inside a row, every pair placed exactly once; the reference's
    own list for KITTI 06 (tests/golden/pair_lists_3_20.npz) gives 1101 row graphs."""
    import ctypes
    from sg_pr_amd import engine

    class Host:            # the plan needs the library, not a GPU
        lib, device = engine.load_library(), "cpu"

    def check(i1, i2, r, m):
        pl = engine.PairPlan(Host, i1, i2, r, m)
        w, nr, ni, p = pl.host_words, pl.n_rows, pl.n_items, len(i1)
        assert w.size == nr + ni + ni + 1 + 2 * p <= Host.lib.sgpr_pair_plan_ints(p, r)
        row_ids, item_row, item_beg = w[:nr], w[nr:nr + ni], w[nr + ni:nr + 2 * ni + 1]
        cols, pos = w[nr + 2 * ni + 1:nr + 2 * ni + 1 + p], w[nr + 2 * ni + 1 + p:]
        assert np.array_equal(row_ids, np.unique(i1))
        assert item_beg[0] == 0 and item_beg[-1] == p
        assert (np.diff(item_beg) > 0).all() and (np.diff(item_beg) <= 16).all()
        assert ni == sum(-(-c // 16) for c in np.bincount(i1)[np.unique(i1)])
        for it in range(ni):
            ps = pos[item_beg[it]:item_beg[it + 1]]
            assert (np.asarray(i1)[ps] == row_ids[item_row[it]]).all()
            assert np.array_equal(np.asarray(i2)[ps], cols[item_beg[it]:item_beg[it + 1]])
            assert (np.diff(ps) > 0).all()                          # list order kept inside a row graph
        assert np.array_equal(np.sort(pos), np.arange(p))
        return pl

    rng = np.random.default_rng(0)
    check(rng.integers(0, 50, 1000), rng.integers(0, 70, 1000), 50, 70)
    check(np.zeros(33, dtype=np.int64), np.arange(33), 1, 33)       # one row graph, three items
    fx = np.load(os.path.join(REPO, "tests", "golden", "pair_lists_3_20.npz"))
    ij = fx["seq_06"].astype(np.int64)
    pl = check(ij[:, 0], ij[:, 1], 1101, 1101)
    assert pl.n_rows == 1101 and pl.P == 8299
    empty = engine.PairPlan(Host, [], [], 5, 5)
    assert empty.P == 0 and empty.n_rows == 0 and empty.n_items == 0
    for bad in (([5], [0]), ([0], [7]), ([-1], [0])):
        with pytest.raises(engine.SgprError, match="outside"):
            engine.PairPlan(Host, bad[0], bad[1], 5, 7)
    with pytest.raises(ValueError):
        engine.PairPlan(Host, [0, 1], [0], 5, 7)


def test_ragged_blob_round_trip():
    """Engine.ragged_blob / ragged_views: the ragged store as one host buffer (one H2D copy) gives back the three arrays
    of to_ragged, each part 16-byte aligned; an empty store is a valid blob."""
    import numpy as np
    from sg_pr_amd import engine, synth
    c, l, _, _ = synth.kitti_like_sequence(37, 100, 3)
    rc, rl, off = engine.Engine.to_ragged(c, l)
    blob, lay = engine.Engine.ragged_blob(rc, rl, off, pin=False)
    vc, vl, vo = engine.Engine.ragged_views(blob, lay)
    assert np.array_equal(vc.numpy(), rc) and np.array_equal(vl.numpy(), rl) and np.array_equal(vo.numpy(), off)
    assert all(st % 16 == 0 for st, _ in lay.values())
    blob, lay = engine.Engine.ragged_blob(rc[:0], rl[:0], off[:1], pin=False)
    vc, vl, vo = engine.Engine.ragged_views(blob, lay)
    assert vc.shape == (0, 3) and vl.shape == (0,) and vo.tolist() == [0]
