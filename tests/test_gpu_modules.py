"""Stand-alone forms of the reference's building blocks (SURVEY.md 8b "signatures to keep") on the GPU:
dgcnn.knn / dgcnn.get_graph_feature (dgcnn.py:14-49), AttentionModule.forward and TenorNetworkModule.forward
(layers_batch.py:28-39, 70-83) against the reference-generated goldens.  `pytest -m gpu`."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _layer_inputs(g):
    f = g["features"]
    return [f[:, :3, :], g["xyz1"], g["xyz2"], f[:, 3:, :], g["sem1"], g["sem2"]]      # order of golden knn_idx


def test_dgcnn_knn_matches_reference_neighbour_sets(golden_dir):
    from sg_pr_amd import dgcnn
    g = np.load(os.path.join(golden_dir, "kitti3_n100_k10.npz"))
    for li, x in enumerate(_layer_inputs(g)):
        xt = torch.from_numpy(np.ascontiguousarray(x)).cuda()
        idx = dgcnn.knn(xt, 10)
        assert idx.dtype == torch.int64 and tuple(idx.shape) == (x.shape[0], x.shape[2], 10)
        idx = idx.cpu().numpy()
        xr = x.transpose(0, 2, 1).astype(np.float64)                                      # [G, N, C]
        for b in range(x.shape[0]):
            # ties are only ever between feature-identical nodes (padding / same one-hot label): canonicalise them
            canon = np.array([np.flatnonzero((xr[b] == xr[b, j]).all(-1))[0] for j in range(xr.shape[1])])
            mine = np.sort(canon[idx[b]], -1)
            theirs = np.sort(canon[g["knn_idx"][b, li].astype(np.int64)], -1)
            assert (mine == theirs).all(), "layer input %d graph %d" % (li, b)
            # nearest first: distances along the list never decrease (float64 check of the fp32 ranking)
            d = ((xr[b][:, None, :] - xr[b][idx[b]]) ** 2).sum(-1)
            assert (np.diff(d, axis=1) >= -1e-4 * (1 + d[:, 1:])).all()
    # error conventions of the C-ABI surface through the Python wrapper
    from sg_pr_amd.engine import SgprError
    with pytest.raises(SgprError):
        dgcnn.knn(torch.zeros(1, 3, 8, device="cuda"), 9)          # k > N
    with pytest.raises(RuntimeError):
        dgcnn.knn(torch.zeros(1, 3, 8), 2)                          # CPU tensor: no fallback


def test_get_graph_feature_is_the_exact_gather(golden_dir):
    from sg_pr_amd import dgcnn
    g = np.load(os.path.join(golden_dir, "kitti3_n100_k10.npz"))
    for li, x in enumerate(_layer_inputs(g)):
        x = np.ascontiguousarray(x)
        idx = g["knn_idx"][:, li].astype(np.int64)                                        # the reference's own lists
        got = dgcnn.get_graph_feature(torch.from_numpy(x).cuda(), k=10, idx=torch.from_numpy(idx).cuda()).cpu().numpy()
        b, c, n = x.shape
        nb = np.stack([x[i][:, idx[i]] for i in range(b)])                                # [B, C, N, k]
        want = np.concatenate((nb - x[:, :, :, None], np.broadcast_to(x[:, :, :, None], nb.shape)), axis=1)
        assert got.shape == (b, 2 * c, n, 10)
        np.testing.assert_array_equal(got, want)                                          # pure gather / subtract: bit-exact
    # idx=None -> own kNN; xyz=True ranks by the first three channels (dgcnn.py:28-31)
    f = torch.from_numpy(np.ascontiguousarray(g["features"])).cuda()
    a = dgcnn.get_graph_feature(f, k=10, xyz=True)
    bq = dgcnn.get_graph_feature(f, k=10, idx=dgcnn.knn(f[:, :3, :].contiguous(), 10))
    assert torch.equal(a, bq) and tuple(a.shape) == (3, 30, 100, 10)


def _args():
    from sg_pr_amd.parser_sg import sgpr_args
    return sgpr_args()


def test_attention_module_forward_stand_alone(golden_dir, ckpt_path, oracle):
    from sg_pr_amd.layers_batch import AttentionModule
    g = np.load(os.path.join(golden_dir, "kitti3_n100_k10.npz"))
    sd = oracle.load_checkpoint(ckpt_path)
    mod = AttentionModule(_args())
    mod.load_state_dict({"weight_matrix": sd["attention.weight_matrix"]})
    mod = mod.cuda().eval()
    rep, scores = mod(torch.from_numpy(g["emb"]).cuda())
    assert tuple(rep.shape) == (3, 32, 1) and tuple(scores.shape) == (3, 100, 1)          # layers_batch.py:38-39
    np.testing.assert_allclose(scores.squeeze(-1).cpu().numpy(), g["att"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(rep.squeeze(-1).cpu().numpy(), g["pooled"], rtol=2e-6, atol=2e-5)


def test_tensor_network_module_forward_stand_alone(golden_dir, ckpt_path, oracle):
    from sg_pr_amd.layers_batch import TenorNetworkModule
    g = np.load(os.path.join(golden_dir, "kitti3_n100_k10.npz"))
    sd = oracle.load_checkpoint(ckpt_path)
    mod = TenorNetworkModule(_args())
    mod.load_state_dict({k: sd["tensor_network." + k] for k in ("weight_matrix", "weight_matrix_block", "bias")})
    mod = mod.cuda().eval()
    pooled = torch.from_numpy(g["pooled"]).cuda().unsqueeze(-1)                           # [3, 32, 1]
    i, j = g["pair_ij"][:, 0].astype(np.int64), g["pair_ij"][:, 1].astype(np.int64)
    out = mod(pooled[i], pooled[j])
    assert tuple(out.shape) == (9, 16, 1)                                                 # layers_batch.py:82-83
    np.testing.assert_allclose(out.squeeze(-1).cpu().numpy(), g["ntn"], rtol=1e-5, atol=2e-5)
    a, b = out[2].clone(), mod(pooled[j[2:3]], pooled[i[2:3]])[0]                         # asymmetric
    assert not torch.equal(a, b)
