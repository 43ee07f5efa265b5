"""Per-head RMSNorm used for q_norm / k_norm — reference: layers/RMSNorm.py:5-33 (weight
multiply in fp32 before the down-cast, unlike layers/layernorm.py)."""
from __future__ import annotations

import torch
from torch import nn

from nanovllm import hip_ops


class RMSNorm(nn.Module):

    def __init__(self, dim: int, eps: float = 1e-6):
        super().__init__()
        self.dim = dim
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim), requires_grad=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dim() == 3:
            return hip_ops.head_rmsnorm(x, self.weight.data, self.eps)
        shape = x.shape
        y = hip_ops.head_rmsnorm(x.reshape(-1, 1, shape[-1]), self.weight.data, self.eps)
        return y.view(shape)
