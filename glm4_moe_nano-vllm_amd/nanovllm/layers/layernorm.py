"""RMSNorm with optional fused residual add — reference: layers/layernorm.py:6-50.  The device
work is glm_rmsnorm / glm_add_rmsnorm (one HIP kernel each; the reference runs 5–7 eager ops)."""
from __future__ import annotations

import torch
from torch import nn

from nanovllm import hip_ops


class RMSNorm(nn.Module):

    def __init__(self, hidden_size: int, eps: float = 1e-6) -> None:
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(hidden_size))

    def rms_forward(self, x) -> torch.Tensor:
        if isinstance(x, hip_ops.DeferredCombine):
            x = x.materialize()
        return hip_ops.rmsnorm(x, self.weight.data, self.eps)

    def add_rms_forward(self, x, residual: torch.Tensor):
        if isinstance(x, hip_ops.PartialSum):  # deferred K-slice reduction of the producing linear layer
            return hip_ops.add_rmsnorm_partial(x, residual, self.weight.data, self.eps)
        if isinstance(x, hip_ops.DeferredCombine):  # the previous layer's MoE combine, done in this launch
            return hip_ops.moe_combine_add_rmsnorm(x, residual, self.weight.data, self.eps)
        return hip_ops.add_rmsnorm(x, residual, self.weight.data, self.eps)

    def forward(self, x: torch.Tensor, residual: torch.Tensor | None = None):
        if residual is None:
            return self.rms_forward(x)
        return self.add_rms_forward(x, residual)
