"""SiluAndMul — reference: layers/activation.py:6-14 (glm_silu_and_mul).  Inside the MLP/MoE
blocks the activation is fused into the gate|up GEMM epilogue instead."""
from __future__ import annotations

import torch
from torch import nn

from nanovllm import hip_ops


class SiluAndMul(nn.Module):

    def __init__(self):
        super().__init__()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return hip_ops.silu_and_mul(x)
