"""Paged-KV attention operator — reference: layers/attention.py:33-112.

Device work: glm_kv_store (the reference's Triton store_kvcache_kernel), glm_paged_attn_decode
(flash_attn_with_kvcache) and glm_varlen_attn_prefill (flash_attn_varlen_func).  Metadata
arrives through the module-global Context exactly as in the reference."""
from __future__ import annotations

import torch
from torch import nn

from nanovllm import hip_ops
from nanovllm.utils.context import get_context


def store_kvcache(key: torch.Tensor, value: torch.Tensor, k_cache: torch.Tensor,
                  v_cache: torch.Tensor, slot_mapping: torch.Tensor):
    N, num_heads, head_dim = key.shape
    D = num_heads * head_dim
    assert key.stride(-1) == 1 and value.stride(-1) == 1
    assert key.stride(1) == head_dim and value.stride(1) == head_dim
    assert k_cache.stride(1) == D and v_cache.stride(1) == D
    assert slot_mapping.numel() == N
    hip_ops.kv_store(key, value, k_cache, v_cache, slot_mapping)


class Attention(nn.Module):

    def __init__(self, num_heads, head_dim, scale, num_kv_heads, cache_config=None,
                 quant_config=None, prefix=None):
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = head_dim
        self.scale = scale
        self.num_kv_heads = num_kv_heads
        self.k_cache = self.v_cache = torch.tensor([])

    def attend(self, q: torch.Tensor, k: torch.Tensor | None, v: torch.Tensor | None) -> torch.Tensor:
        """Attention proper, K/V of the current tokens already in the cache (or passed for the
        cache-less prefill)."""
        context = get_context()
        k_cache, v_cache = self.k_cache, self.v_cache
        if context.is_prefill:
            block_size = k_cache.shape[1] if k_cache.numel() else 256
            if context.block_tables is not None:  # prefix cache: K/V come from the paged cache
                return hip_ops.varlen_attn_prefill(q, k_cache, v_cache, context.cu_seqlens_q,
                                                   context.cu_seqlens_k, context.max_seqlen_q,
                                                   self.scale, context.block_tables, block_size)
            return hip_ops.varlen_attn_prefill(q, k, v, context.cu_seqlens_q, context.cu_seqlens_k,
                                               context.max_seqlen_q, self.scale)
        return hip_ops.paged_attn_decode(q, k_cache, v_cache, context.context_lens,
                                         context.block_tables, self.scale)

    def forward(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
        context = get_context()
        if self.k_cache.numel() and self.v_cache.numel():
            store_kvcache(k, v, self.k_cache, self.v_cache, context.slot_mapping)
        o = self.attend(q, k, v)
        # the reference's decode branch returns [B, 1, nh, hd] (q.unsqueeze(1) in, flash-attn out)
        return o if context.is_prefill else o.unsqueeze(1)
