#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE itself.

Run only in the build container (needs /root/reference; the GPU box has no copy):

    python tests/golden/make_golden.py

The reference (kxhit/SG_PR) is imported from /root/reference on CPU with the
shims listed in SURVEY.md §8(c): stub modules for the absent `tensorboardX` /
`texttable`, `yaml.load` given a Loader, `.cuda()` made the identity,
`torch.load` forced to map_location='cpu', `torch.arange` stripped of its
`device=` kwarg.  Nothing from the reference's sources is copied: the outputs
are data (inputs + expected outputs) stored as small .npz files.

Fixtures written (all float32 unless noted):
  kitti3_n100_k10.npz   the three shipped graphs at the shipped config
                        (node_num=100, K=10): packed features, every
                        intermediate of SG.forward, the 9 ordered-pair scores,
                        process_pair distances and eval_batch_pair (pred, gt)
  synth_n64_k10.npz / synth_n100_k10.npz / synth_n256_k20.npz
                        seeded synthetic pairs (sg_pr_amd.synth) with pooled
                        vectors, attention and scores
  prf1.npz              random (score, gt) vectors with the F1-max computed the
                        way eval_batch.py:69-87 does (sklearn PR curve), plus the
                        ROC curve / AUC of eval_batch.py:48-49
  edge_n100_k10.npz     graphs at the edges of the tie regime: k-1 and
                        k padded slots, a single label, fewer than 17 nodes, no padding
                        with >= k nodes per label, trailing duplicate real nodes
  release_models.npz    scores of the reference under each of the 18 checkpoints of
                        model/release_model.zip (the zip itself is copied beside it as
                        a data fixture: it holds weights only): the 9 shipped pairs
                        and one synthetic batch per checkpoint
"""
import os
import sys
import types
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)


def import_reference():
    import yaml
    import torch
    import matplotlib
    matplotlib.use("Agg")

    tbx = types.ModuleType("tensorboardX")

    class SummaryWriter:  # no-op stand-in for the absent tensorboardX
        def __init__(self, *a, **k):
            pass

        def add_scalar(self, *a, **k):
            pass

    tbx.SummaryWriter = SummaryWriter
    sys.modules["tensorboardX"] = tbx

    tt = types.ModuleType("texttable")

    class Texttable:
        def add_rows(self, rows):
            self.rows = rows

        def draw(self):
            return ""

    tt.Texttable = Texttable
    sys.modules["texttable"] = tt

    _yaml_load = yaml.load
    yaml.load = lambda s, Loader=None: _yaml_load(s, Loader=yaml.FullLoader)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    _torch_load = torch.load
    torch.load = lambda f, map_location=None, **k: _torch_load(f, map_location="cpu", **k)
    _arange = torch.arange

    def arange(*a, **k):
        k.pop("device", None)
        return _arange(*a, **k)

    torch.arange = arange

    sys.path.insert(0, REF)
    import sg_net  # noqa: E402  (the reference)
    import dgcnn   # noqa: E402
    import utils as ref_utils  # noqa: E402
    import parser_sg  # noqa: E402
    return sg_net, dgcnn, ref_utils, parser_sg


def make_args(parser_sg, node_num, k, logdir):
    args = parser_sg.sgpr_args()
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        args.load("./config/config.yml")
    finally:
        os.chdir(cwd)
    args.model = os.path.join(REF, "model", "model.pth")
    args.node_num = node_num
    args.K = k
    args.logdir = logdir
    return args


def step_by_step(model, dgcnn, feats, k):
    """SG.dgcnn_conv_pass (sg_net.py:79-110) executed call by call to expose
    every intermediate; the caller checks the result against the real method."""
    import torch
    out = {}
    xyz = feats[:, :3, :]
    sem = feats[:, 3:, :]
    idx_all = []

    def edge(x, block, name):
        idx = dgcnn.knn(x, k=k)
        idx_all.append(idx.numpy().astype(np.int16))
        y = dgcnn.get_graph_feature(x, k=k, cuda=0)
        y = block(y).max(dim=-1, keepdim=False)[0]
        out[name] = y.detach().numpy()
        return y

    with torch.no_grad():
        x1 = edge(xyz, model.dgcnn_s_conv1, "xyz1")
        x2 = edge(x1, model.dgcnn_s_conv2, "xyz2")
        x3 = edge(x2, model.dgcnn_s_conv3, "xyz3")
        s1 = edge(sem, model.dgcnn_f_conv1, "sem1")
        s2 = edge(s1, model.dgcnn_f_conv2, "sem2")
        s3 = edge(s2, model.dgcnn_f_conv3, "sem3")
        e = model.dgcnn_conv_end(torch.cat((x3, s3), dim=1)).permute(0, 2, 1)
    out["knn_idx"] = np.stack(idx_all, axis=1)  # [B, 6, N, k] order: xyz1..3, sem1..3 inputs
    out["emb"] = e.numpy()
    return out


def main():
    import torch
    torch.manual_seed(0)
    torch.set_num_threads(8)
    sg_net, dgcnn, ref_utils, parser_sg = import_reference()
    from sg_pr_amd import synth
    tmp = tempfile.mkdtemp(prefix="sgpr_golden_")

    # ---------------------------------------------------------------- shipped graphs
    args = make_args(parser_sg, 100, 10, tmp)
    trainer = sg_net.SGTrainer(args, False)
    trainer.model.eval()
    model = trainer.model.module if hasattr(trainer.model, "module") else trainer.model
    names = ["0", "3", "250"]
    paths = [os.path.join(REF, "data", n + ".json") for n in names]

    feats = []
    for p in paths:
        d = ref_utils.process_pair([p, p])
        t = trainer.transfer_to_torch(d, False)
        feats.append(np.asarray(t["features_1"], dtype=np.float64))
    feats64 = np.stack(feats)
    feats_t = torch.FloatTensor(feats64)

    g = step_by_step(model, dgcnn, feats_t, 10)
    with torch.no_grad():
        e_real = model.dgcnn_conv_pass(feats_t)
        assert torch.equal(e_real, torch.from_numpy(g["emb"])), "step-by-step != dgcnn_conv_pass"
        pooled, att = model.attention(e_real)
    g["features"] = feats_t.numpy()
    g["pooled"] = pooled.numpy().reshape(3, -1)
    g["att"] = att.numpy().reshape(3, -1)

    pair_ij, scores, ntn, dist, pred_eb, gt_eb = [], [], [], [], [], []
    for i in range(3):
        for j in range(3):
            pair_ij.append((i, j))
            data = {"features_1": feats_t[i:i + 1], "features_2": feats_t[j:j + 1]}
            with torch.no_grad():
                s, a1, a2 = trainer.model(data)
                t = model.tensor_network(pooled[i:i + 1], pooled[j:j + 1])
            scores.append(float(s[0]))
            ntn.append(t.numpy().reshape(-1))
            dist.append(ref_utils.process_pair([paths[i], paths[j]])["distance"])
            p, gt = trainer.eval_batch_pair([[paths[i], paths[j]]])
            pred_eb.append(p[0])
            gt_eb.append(gt[0])
    # batched call (all nine pairs at once) for the batch-invariance check
    data = {"features_1": torch.stack([feats_t[i] for i, _ in pair_ij]),
            "features_2": torch.stack([feats_t[j] for _, j in pair_ij])}
    with torch.no_grad():
        s_b, _, _ = trainer.model(data)
    np.savez_compressed(
        os.path.join(HERE, "kitti3_n100_k10.npz"),
        names=np.array(names), pair_ij=np.array(pair_ij, dtype=np.int32),
        scores=np.array(scores, dtype=np.float32), scores_batched=s_b.numpy(),
        ntn=np.stack(ntn).astype(np.float32), distance=np.array(dist, dtype=np.float64),
        eval_batch_pred=np.array(pred_eb, dtype=np.float32), eval_batch_gt=np.array(gt_eb, dtype=np.float64),
        **{k: v for k, v in g.items()})
    print("kitti3: scores", np.array(scores))

    # ---------------------------------------------------------------- synthetic
    def synth_case(fname, node_num, k, lo, hi, pairs, seed):
        a = make_args(parser_sg, node_num, k, tmp)
        tr = sg_net.SGTrainer(a, False)
        tr.model.eval()
        m = tr.model.module if hasattr(tr.model, "module") else tr.model
        centers, labels, n_real = synth.make_graphs(2 * pairs, node_num, lo, hi, seed)
        dense = torch.from_numpy(synth.dense_features(centers, labels))
        with torch.no_grad():
            s, a1, a2 = tr.model({"features_1": dense[0::2], "features_2": dense[1::2]})
            e = m.dgcnn_conv_pass(dense)
            pl, at = m.attention(e)
        np.savez_compressed(os.path.join(HERE, fname), centers=centers, labels=labels, n_real=n_real,
                            node_num=node_num, k=k, seed=seed,
                            scores=s.numpy(), pooled=pl.numpy().reshape(2 * pairs, -1),
                            att=at.numpy().reshape(2 * pairs, -1),
                            emb_sum=e.numpy().sum(axis=(1, 2)))
        print(fname, "scores[:4]", s.numpy()[:4])

    synth_case("synth_n64_k10.npz", 64, 10, 20, 54, 16, 0)
    synth_case("synth_n100_k10.npz", 100, 10, 25, 60, 8, 1)
    synth_case("synth_n256_k20.npz", 256, 20, 100, 236, 4, 2)

    # ---------------------------------------------------------------- edges of the tie regime
    def edge_graphs(node_num=100, k=10):
        rng = np.random.default_rng(17)
        centers = np.zeros((12, node_num, 3), dtype=np.float32)
        labels = -np.ones((12, node_num), dtype=np.int32)

        def fill(g, n, lab):
            centers[g, :n, :2] = rng.uniform(-50, 50, size=(n, 2))
            centers[g, :n, 2] = rng.uniform(-2, 1, size=n)
            labels[g, :n] = np.sort(np.asarray(lab))

        fill(0, node_num - k + 1, rng.integers(0, 12, node_num - k + 1))     # k-1 pads: every copy kept as a slot
        fill(1, node_num - k, rng.integers(0, 12, node_num - k))             # exactly k pads: one representative
        fill(2, 40, np.full(40, 3))                                          # a single label
        fill(3, 12, rng.integers(0, 12, 12))                                 # < 17 slots processed
        fill(4, node_num, np.repeat([0, 5, 7, 9], node_num // 4))            # no padding, >= k nodes per label
        fill(5, 60, rng.integers(0, 12, 60))                                 # trailing duplicate REAL nodes + pads
        centers[5, 48:60] = centers[5, 59]
        labels[5, 48:60] = labels[5, 59]
        fill(6, node_num, np.repeat([1, 2, 3, 4], node_num // 4))            # no padding, last 12 real nodes identical
        centers[6, 88:] = centers[6, 99]
        fill(7, 1, [6])                                                      # one node
        fill(8, node_num - k + 1, np.full(node_num - k + 1, 11))             # k-1 pads and a single label
        fill(9, 17, rng.integers(0, 12, 17))                                 # 17 nodes + representative = 18 slots
        fill(10, 50, np.repeat(np.arange(10), 5))                            # 5 nodes per label: every row needs pads
        fill(11, 33, rng.integers(0, 3, 33))
        return centers, labels

    a = make_args(parser_sg, 100, 10, tmp)
    tr = sg_net.SGTrainer(a, False)
    tr.model.eval()
    m = tr.model.module if hasattr(tr.model, "module") else tr.model
    ec, el = edge_graphs()
    dense = torch.from_numpy(synth.dense_features(ec, el))
    with torch.no_grad():
        e = m.dgcnn_conv_pass(dense)
        pl, at = m.attention(e)
        ii, jj = np.meshgrid(np.arange(12), np.arange(12), indexing="ij")
        s, _, _ = tr.model({"features_1": dense[ii.reshape(-1)], "features_2": dense[jj.reshape(-1)]})
    np.savez_compressed(os.path.join(HERE, "edge_n100_k10.npz"), centers=ec, labels=el, node_num=100, k=10,
                        pooled=pl.numpy().reshape(12, -1), att=at.numpy().reshape(12, -1), emb=e.numpy(),
                        scores=s.numpy().reshape(12, 12))
    print("edge: score matrix diag", np.diag(s.numpy().reshape(12, 12)))

    # ---------------------------------------------------------------- the 18 checkpoints of release_model.zip
    import io
    import shutil
    import zipfile
    zsrc = os.path.join(REF, "model", "release_model.zip")
    shutil.copyfile(zsrc, os.path.join(HERE, "release_model.zip"))           # weights only: a data fixture
    centers, labels, _ = synth.make_graphs(16, 100, 25, 60, 21, kitti_like=True)
    dsyn = torch.from_numpy(synth.dense_features(centers, labels))
    out = {"names": [], "pair_ij": np.array(pair_ij, dtype=np.int32), "syn_centers": centers, "syn_labels": labels}
    with zipfile.ZipFile(zsrc) as z:
        for name in sorted(n for n in z.namelist() if n.endswith("model.pth")):
            sd = torch.load(io.BytesIO(z.read(name)), map_location="cpu")
            m.load_state_dict({k[7:] if k.startswith("module.") else k: v for k, v in sd.items()})
            m.eval()
            with torch.no_grad():
                s9, _, _ = m(data)
                ssyn, _, _ = m({"features_1": dsyn[0::2], "features_2": dsyn[1::2]})
            key = name.split("/", 1)[1].rsplit("/", 1)[0]                     # e.g. 3_20/00
            out["names"].append(key)
            out["scores9/" + key] = s9.numpy()
            out["scores_syn/" + key] = ssyn.numpy()
            print("release", key, "scores9[:3]", s9.numpy()[:3])
    out["names"] = np.array(out["names"])
    np.savez_compressed(os.path.join(HERE, "release_models.npz"), **out)

    # ---------------------------------------------------------------- PR / F1-max (eval_batch.py:69-87)
    from sklearn import metrics
    rng = np.random.default_rng(3)
    cases = {}
    for c, (n, pos_rate, quant) in enumerate([(1000, 0.05, None), (5000, 0.01, None), (2000, 0.2, 64), (50, 0.5, 4)]):
        gt = (rng.random(n) < pos_rate).astype(np.float64)
        gt[0] = 1.0
        sc = np.clip(0.6 * gt + rng.normal(0.2, 0.25, size=n), 0, 1).astype(np.float32)
        if quant:
            sc = (np.round(sc * quant) / quant).astype(np.float32)  # heavy ties
        precision, recall, thr = metrics.precision_recall_curve(gt, sc)
        with np.errstate(divide="ignore", invalid="ignore"):
            f1 = 2 * precision * recall / (precision + recall)
        f1 = np.nan_to_num(f1)
        fpr, tpr, roc_thr = metrics.roc_curve(gt, sc)
        cases[f"fpr{c}"] = fpr
        cases[f"tpr{c}"] = tpr
        cases[f"roc_thr{c}"] = roc_thr
        cases[f"gt{c}"] = gt
        cases[f"score{c}"] = sc
        cases[f"f1max{c}"] = np.float64(np.max(f1))
        cases[f"precision{c}"] = precision
        cases[f"recall{c}"] = recall
        cases[f"auc{c}"] = np.float64(metrics.auc(fpr, tpr))
    cases["ncases"] = 4
    np.savez_compressed(os.path.join(HERE, "prf1.npz"), **cases)
    print("prf1 f1max:", [float(cases[f"f1max{c}"]) for c in range(4)])


if __name__ == "__main__":
    main()
