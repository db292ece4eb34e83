"""`dgcnn.knn` / `dgcnn.get_graph_feature` with the reference's signatures (dgcnn.py:14-20, 23-49), backed by the
HIP kernels behind `sgpr_knn` / `sgpr_graph_feature` (include/sgpr.h).

Inside `SG.forward` neither symbol is called: the fused embed kernel selects neighbours from an MFMA Gram matrix and
never builds the `[B, 2C, N, k]` edge tensor.  These stand-alone forms serve callers that use the two
functions on their own tensors; GPU tensors only (no CPU fallback).

Tie order: `torch.topk` leaves the order of equal distances implementation-defined (its CPU and CUDA kernels
disagree); `knn` here returns equal-distance candidates lowest index first.  With >= k-1 padded slots per graph the
choice does not change any EdgeConv output.
"""
import torch

from . import engine as _engine


def knn(x, k):
    """x [B, C, N] -> idx [B, N, k] int64: the k nearest nodes of every node (itself included), nearest first."""
    return _engine.knn(x, int(k))


def get_graph_feature(x, k=20, cuda=0, idx=None, xyz=False):
    """x [B, C, N] -> edge features [B, 2C, N, k] = cat(x_j - x_i, x_i) over the k neighbours of every node.

    `cuda` (the reference builds `torch.device('cuda:' + str(cuda))` from it, dgcnn.py:32) is accepted and ignored:
    the result lives on x's device.  `idx`: precomputed neighbour lists; `xyz=True` ranks by the first three channels."""
    batch_size = x.size(0)
    num_points = x.size(2)
    x = x.reshape(batch_size, -1, num_points)
    if idx is None:
        idx = knn(x[:, :3, :] if xyz else x, k=k)
    return _engine.graph_feature(x, idx)
