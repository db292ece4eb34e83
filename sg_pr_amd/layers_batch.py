"""`AttentionModule` / `TenorNetworkModule` with the reference's names, state-dict keys and forward contracts
(layers_batch.py:3-83).  Inside `sg_net.SG` their maths runs fused in the HIP engine (attention in the tail of
embed_kernel, the NTN in the score kernels); called stand-alone, `forward` runs the dedicated HIP kernels behind
`sgpr_attention_pool` / `sgpr_ntn` (include/sgpr.h) with the module's own parameters.  Inference only, GPU tensors
only: there is no CPU fallback and no autograd through the kernels.
"""
import torch

from . import engine as _engine


class AttentionModule(torch.nn.Module):
    """layers_batch.py:3-39 - `weight_matrix` [F3, F3], xavier-uniform."""

    def __init__(self, args):
        super(AttentionModule, self).__init__()
        self.args = args
        self.weight_matrix = torch.nn.Parameter(torch.Tensor(self.args.filters_3, self.args.filters_3))
        torch.nn.init.xavier_uniform_(self.weight_matrix)

    def forward(self, embedding):
        """layers_batch.py:28-39 - embedding [B, N, F3] -> (representation [B, F3, 1], sigmoid_scores [B, N, 1])."""
        rep, att = _engine.attention_pool(self.weight_matrix, embedding)
        return rep.unsqueeze(-1), att.unsqueeze(-1)


class TenorNetworkModule(torch.nn.Module):
    """layers_batch.py:41-83 (the reference's spelling is part of its API)."""

    def __init__(self, args):
        super(TenorNetworkModule, self).__init__()
        self.args = args
        f3, t = self.args.filters_3, self.args.tensor_neurons
        self.weight_matrix = torch.nn.Parameter(torch.Tensor(f3, f3, t))
        self.weight_matrix_block = torch.nn.Parameter(torch.Tensor(t, 2 * f3))
        self.bias = torch.nn.Parameter(torch.Tensor(t, 1))
        torch.nn.init.xavier_uniform_(self.weight_matrix)
        torch.nn.init.xavier_uniform_(self.weight_matrix_block)
        torch.nn.init.xavier_uniform_(self.bias)

    def forward(self, embedding_1, embedding_2):
        """layers_batch.py:70-83 - embedding_1/2 [B, F3, 1] -> scores [B, T, 1]."""
        b = embedding_1.shape[0]
        out = _engine.ntn(self.weight_matrix, self.weight_matrix_block, self.bias,
                          embedding_1.reshape(b, -1), embedding_2.reshape(b, -1))
        return out.unsqueeze(-1)
