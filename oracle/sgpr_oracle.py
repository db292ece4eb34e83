"""CPU ORACLE for the SG_PR pair-scoring hot path.  TEST INFRASTRUCTURE ONLY.

This file is the checker, not the product: only `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s `cpu_baseline` leg may import it.  Nothing under `sg_pr_amd/`
imports it and the product path raises when the HIP library is missing.

It is a from-scratch functional restatement (torch CPU, fp32 by default) of the
reference algorithm, written from SURVEY.md §3.3 and citing the reference
file:line each function follows.  It keeps the reference's *formulation* (dense
pairwise distance by expansion, `topk`, materialised `[x_j - x_i, x_i]` edge
tensor, 1x1 conv -> eval BatchNorm -> LeakyReLU(0.2) -> max over k), i.e. it is
also the honest "what the reference does on CPU" cost model for bench.py.

Parity pin: tests/test_oracle_golden.py checks every function here against the
golden vectors in tests/golden/*.npz, which were produced by running the
reference itself (tests/golden/make_golden.py) - scores, every intermediate of
SG.forward for the three shipped graphs, eval_batch_pair (pred, gt), F1-max.
"""
import json
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

NUM_LABELS = 12
BN_EPS = 1e-5          # torch.nn.BatchNorm default, used unchanged by sg_net.py:50-76
LRELU_SLOPE = 0.2      # sg_net.py:53


# --------------------------------------------------------------------------- checkpoint
def load_checkpoint(path):
    """sg_net.py:164-174 - load a DataParallel state dict and strip `module.`."""
    sd = torch.load(path, map_location="cpu")
    out = OrderedDict()
    for k, v in sd.items():
        out[k[7:] if k.startswith("module.") else k] = v
    return out


# --------------------------------------------------------------------------- host packing
def read_graph(path):
    with open(path) as f:
        return json.load(f)


def process_pair(paths):
    """utils.py:21-38 - two graph JSONs -> dict with planar pose distance."""
    d1, d2 = read_graph(paths[0]), read_graph(paths[1])
    p1, p2 = d1["pose"], d2["pose"]
    return {
        "centers_1": d1["centers"], "nodes_1": d1["nodes"],
        "centers_2": d2["centers"], "nodes_2": d2["nodes"],
        "distance": math.sqrt((p1[3] - p2[3]) ** 2 + (p1[11] - p2[11]) ** 2),
    }


def pack_graph(centers, nodes, node_num, num_labels=NUM_LABELS):
    """sg_net.py:250-299 (one side) - pad with label -1 / centre 0, one-hot,
    concat xyz | one-hot, transpose -> float64 [(3+L), node_num].
    Graphs larger than node_num are rejected (the reference subsamples them with
    an unseeded np.random.choice, sg_net.py:252-256: no parity definable)."""
    n = len(nodes)
    if n > node_num:
        raise ValueError("graph has %d nodes > node_num=%d (reference subsamples randomly)" % (n, node_num))
    c = np.zeros((node_num, 3), dtype=np.float64)
    c[:n] = np.asarray(centers, dtype=np.float64).reshape(n, 3)
    onehot = np.zeros((node_num, num_labels), dtype=np.float64)
    for i, lab in enumerate(nodes):
        lab = int(lab)
        if lab < 0 or lab >= num_labels:
            raise KeyError(lab)  # sg_net.py:277 global_labels[node]
        onehot[i, lab] = 1.0
    return np.concatenate((c, onehot), axis=1).T


def target_from_distance(distance, p_thresh):
    """sg_net.py:302-309 - 1 if d<=p_thresh, 0 if d>=20, else the reference exits."""
    if distance <= p_thresh:
        return 1.0
    if distance >= 20:
        return 0.0
    raise SystemExit("distance error: %r" % distance)


# --------------------------------------------------------------------------- DGCNN ops
def knn(x, k):
    """dgcnn.py:14-20 - x [B,C,N] -> idx [B,N,k] of the k largest -||xi-xj||^2,
    distance by expansion, self included."""
    return neg_sq_dist(x).topk(k=k, dim=-1)[1]


def neg_sq_dist(x):
    """dgcnn.py:15-17 - x [B,C,N] -> pd [B,N,N] = -||xi-xj||^2 by expansion, in exactly the reference's fp32 operations."""
    inner = -2 * torch.matmul(x.transpose(2, 1), x)
    xx = torch.sum(x ** 2, dim=1, keepdim=True)
    return -xx - inner - xx.transpose(2, 1)


def graph_feature(x, k, idx=None):
    """dgcnn.py:23-49 - [B,C,N] -> edge tensor [B,2C,N,k] = cat(x_j - x_i, x_i)."""
    b, c, n = x.shape
    if idx is None:
        idx = knn(x, k)
    xt = x.transpose(2, 1).contiguous()                       # [B,N,C]
    flat = (idx + torch.arange(b).view(-1, 1, 1) * n).view(-1)
    nb = xt.view(b * n, c)[flat].view(b, n, k, c)
    ctr = xt.view(b, n, 1, c).expand(b, n, k, c)
    return torch.cat((nb - ctr, ctr), dim=3).permute(0, 3, 1, 2)


def _bn(z, sd, prefix):
    return F.batch_norm(z, sd[prefix + ".1.running_mean"], sd[prefix + ".1.running_var"],
                        sd[prefix + ".1.weight"], sd[prefix + ".1.bias"], False, 0.0, BN_EPS)


def edgeconv(x, sd, prefix, k):
    """sg_net.py:50-73 + :85-102 - Conv2d 1x1 (no bias) -> BN(eval) -> LeakyReLU -> max_k."""
    z = F.conv2d(graph_feature(x, k), sd[prefix + ".0.weight"])
    z = F.leaky_relu(_bn(z, sd, prefix), LRELU_SLOPE)
    return z.max(dim=-1)[0]


def conv_pass(sd, feats, k, want_layers=False):
    """SG.dgcnn_conv_pass, sg_net.py:79-110 - [B,3+L,N] -> [B,N,F3]."""
    xyz, sem = feats[:, :3, :], feats[:, 3:, :]
    x1 = edgeconv(xyz, sd, "dgcnn_s_conv1", k)
    x2 = edgeconv(x1, sd, "dgcnn_s_conv2", k)
    x3 = edgeconv(x2, sd, "dgcnn_s_conv3", k)
    s1 = edgeconv(sem, sd, "dgcnn_f_conv1", k)
    s2 = edgeconv(s1, sd, "dgcnn_f_conv2", k)
    s3 = edgeconv(s2, sd, "dgcnn_f_conv3", k)
    z = F.conv1d(torch.cat((x3, s3), dim=1), sd["dgcnn_conv_end.0.weight"])
    e = F.leaky_relu(_bn(z, sd, "dgcnn_conv_end"), LRELU_SLOPE).permute(0, 2, 1)
    if want_layers:
        return e, {"xyz1": x1, "xyz2": x2, "xyz3": x3, "sem1": s1, "sem2": s2, "sem3": s3}
    return e


# --------------------------------------------------------------------------- SimGNN tail
def attention(sd, emb):
    """AttentionModule.forward, layers_batch.py:28-39 - no pad mask, divisor N."""
    b = emb.shape[0]
    ctx = torch.tanh(torch.mean(torch.matmul(emb, sd["attention.weight_matrix"]), dim=1))
    sig = torch.sigmoid(torch.matmul(emb, ctx.view(b, -1, 1)))
    rep = torch.matmul(emb.permute(0, 2, 1), sig)
    return rep, sig                                            # [B,F,1], [B,N,1]


def tensor_network(sd, e1, e2):
    """TenorNetworkModule.forward, layers_batch.py:70-83."""
    b, f = e1.shape[0], e1.shape[1]
    w = sd["tensor_network.weight_matrix"]
    t = w.shape[2]
    s = torch.matmul(e1.permute(0, 2, 1), w.view(f, -1)).view(b, f, t)
    s = torch.matmul(s.permute(0, 2, 1), e2)
    blk = torch.matmul(sd["tensor_network.weight_matrix_block"], torch.cat((e1, e2), dim=1))
    return F.relu(s + blk + sd["tensor_network.bias"])


def head(sd, ntn):
    """sg_net.py:131-136."""
    h = F.relu(F.linear(ntn.permute(0, 2, 1), sd["fully_connected_first.weight"], sd["fully_connected_first.bias"]))
    return torch.sigmoid(F.linear(h, sd["scoring_layer.weight"], sd["scoring_layer.bias"])).reshape(-1)


def forward(sd, features_1, features_2, k):
    """SG.forward, sg_net.py:112-138 -> (score[B], att1[B,N,1], att2[B,N,1])."""
    with torch.no_grad():
        e1 = conv_pass(sd, features_1, k)
        e2 = conv_pass(sd, features_2, k)
        p1, a1 = attention(sd, e1)
        p2, a2 = attention(sd, e2)
        return head(sd, tensor_network(sd, p1, p2)), a1, a2


def embed(sd, feats, k):
    """Per-graph half of forward: [G,3+L,N] -> (pooled [G,F], att [G,N], emb [G,N,F])."""
    with torch.no_grad():
        e = conv_pass(sd, feats, k)
        p, a = attention(sd, e)
    return p.reshape(p.shape[0], -1), a.reshape(a.shape[0], -1), e


def score_from_pooled(sd, p1, p2):
    """Pair-coupled half of forward on pooled vectors [B,F] x [B,F] -> score [B]."""
    with torch.no_grad():
        return head(sd, tensor_network(sd, p1.unsqueeze(-1), p2.unsqueeze(-1)))


def score_all_pairs(sd, pooled_rows, pooled_cols, chunk=256):
    """Dense score matrix [R,M] by evaluating every ordered pair the faithful way."""
    r, m = pooled_rows.shape[0], pooled_cols.shape[0]
    out = torch.empty(r, m)
    for i0 in range(0, r, chunk):
        rows = pooled_rows[i0:i0 + chunk]
        a = rows.repeat_interleave(m, dim=0)
        b = pooled_cols.repeat(rows.shape[0], 1)
        out[i0:i0 + chunk] = score_from_pooled(sd, a, b).view(rows.shape[0], m)
    return out


def eval_batch_pair(sd, batch, node_num, k, p_thresh):
    """SGTrainer.eval_batch_pair, sg_net.py:503-525 -> (pred f32 [B], gt f64 [B])."""
    f1, f2, gt = [], [], []
    for pair in batch:
        d = process_pair(pair)
        f1.append(pack_graph(d["centers_1"], d["nodes_1"], node_num))
        f2.append(pack_graph(d["centers_2"], d["nodes_2"], node_num))
        gt.append(target_from_distance(d["distance"], p_thresh))
    s, _, _ = forward(sd, torch.FloatTensor(np.array(f1)), torch.FloatTensor(np.array(f2)), k)
    return s.numpy().reshape(-1), np.array(gt).reshape(-1)


# --------------------------------------------------------------------------- metrics
def precision_recall_curve(gt, score):
    """Restatement of sklearn.metrics.precision_recall_curve as used by
    eval_batch.py:69 (sklearn >= 1.1 semantics: no truncation at full recall;
    one point per distinct threshold, ascending thresholds, final (P=1, R=0))."""
    gt = np.asarray(gt, dtype=np.float64).ravel()
    score = np.asarray(score).ravel()
    order = np.argsort(score, kind="mergesort")[::-1]
    s, y = score[order], gt[order]
    distinct = np.where(np.diff(s))[0]
    ends = np.r_[distinct, y.size - 1]
    tps = np.cumsum(y)[ends]
    fps = 1 + ends - tps
    ps = tps + fps
    precision = np.zeros_like(tps)
    np.divide(tps, ps, out=precision, where=(ps != 0))
    recall = np.ones_like(tps) if tps[-1] == 0 else tps / tps[-1]
    return np.hstack((precision[::-1], 1.0)), np.hstack((recall[::-1], 0.0)), s[ends][::-1]


def f1_max(gt, score):
    """eval_batch.py:85-87 - F1 = 2PR/(P+R), nan_to_num, max."""
    p, r, _ = precision_recall_curve(gt, score)
    with np.errstate(divide="ignore", invalid="ignore"):
        f1 = 2 * p * r / (p + r)
    return float(np.max(np.nan_to_num(f1)))
