"""NeoX rotary embedding on the first rotary_dim dims of each head — reference:
layers/rotary_embedding.py:17-71.  The cos/sin table is built on the host with the same torch
ops as the reference (so the table is bit-identical); the rotation is glm_rope_neox."""
from __future__ import annotations

from functools import lru_cache

import torch
from torch import nn

from nanovllm import hip_ops


class RotaryEmbedding(nn.Module):

    def __init__(self, head_size: int, rotary_dim: int, max_position_embeddings: int,
                 base: float) -> None:
        super().__init__()
        self.head_size = head_size
        self.rotary_dim = rotary_dim
        expo = torch.arange(0, rotary_dim, 2, dtype=torch.float) / rotary_dim
        inv_freq = 1.0 / (base ** expo)
        t = torch.arange(max_position_embeddings, dtype=torch.float)
        freqs = torch.outer(t, inv_freq)
        cache = torch.cat((freqs.cos(), freqs.sin()), dim=-1).unsqueeze(1)
        self.register_buffer("cos_sin_cache", cache, persistent=False)

    def forward(self, positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor):
        return hip_ops.rope_neox(positions, query, key, self.cos_sin_cache, self.rotary_dim)


@lru_cache(1)
def get_rope(head_size: int, rotary_dim: int, max_position: int, base: float,
             rope_scaling: dict | None = None, partial_rotary_factor: float = 1.0):
    assert rope_scaling is None
    if partial_rotary_factor < 1.0:
        rotary_dim = int(head_size * partial_rotary_factor)
    return RotaryEmbedding(head_size, rotary_dim, max_position, base)
