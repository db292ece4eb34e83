"""Parameter containers with the reference's names and state-dict keys
(layers_batch.py:3-83).  Inside `sg_net.SG` their maths runs fused in the HIP
engine; called stand-alone they run the same kernels on pooled / embedded inputs.
"""
import torch


class AttentionModule(torch.nn.Module):
    """layers_batch.py:3-39 - `weight_matrix` [F3, F3], xavier-uniform."""

    def __init__(self, args):
        super(AttentionModule, self).__init__()
        self.args = args
        self.weight_matrix = torch.nn.Parameter(torch.Tensor(self.args.filters_3, self.args.filters_3))
        torch.nn.init.xavier_uniform_(self.weight_matrix)

    def forward(self, embedding):
        raise NotImplementedError(
            "AttentionModule runs fused inside the embed kernel (sg_pr_amd/csrc/sgpr_embed.hip); "
            "call SG.forward / SG.embed, which return the attention scores and pooled vector")


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
        raise NotImplementedError(
            "TenorNetworkModule runs fused with the scoring head (sg_pr_amd/csrc/sgpr_score.hip); "
            "call SG.score_pooled / SG.forward")
