"""GLM-4-MoE attention block — reference: models/glm4_moe/attention_new.py:13-368 (the wrapper
decode_layer.py:5 actually uses; partial rotary 0.5, optional per-head q/k RMSNorm).

forward = qkv GEMM(+bias) → ONE fused kernel (split + q/k norm + rotary + paged-KV scatter) →
paged decode / varlen prefill attention → o GEMM (+ RCCL all-reduce when tp > 1).  The
reference's per-call debug file read (attention_new.py:288-292) is not reproduced."""
from __future__ import annotations

import torch
import torch.nn as nn

from nanovllm import hip_ops
from nanovllm.distributed.parallel_state import get_tp_group
from nanovllm.layers.attention import Attention
from nanovllm.layers.linear import QKVParallelLinear, RowParallelLinear
from nanovllm.layers.RMSNorm import RMSNorm
from nanovllm.layers.rotary_embedding import get_rope
from nanovllm.utils.context import get_context
from nanovllm.utils.loader import load_prefixed


class Glm4MoeAttention(nn.Module):

    def __init__(self, config, prefix: str = "", quant_config: dict | None = None):
        super().__init__()
        if quant_config is not None:
            raise NotImplementedError("AWQ int4 projections are outside the MI355X hot-path scope")
        tp_size = get_tp_group().world_size
        self.config = config
        self.hidden_size = config.hidden_size
        self.total_num_heads = config.num_attention_heads
        self.total_num_kv_heads = config.num_key_value_heads
        if self.total_num_kv_heads >= tp_size:
            assert self.total_num_kv_heads % tp_size == 0
        else:
            assert tp_size % self.total_num_kv_heads == 0
        self.head_dim = config.head_dim
        self.num_heads = self.total_num_heads // tp_size
        self.num_kv_heads = max(1, self.total_num_kv_heads // tp_size)
        self.q_size = self.num_heads * self.head_dim
        self.kv_size = self.num_kv_heads * self.head_dim
        self.scaling = self.head_dim ** -0.5
        self.qkv_bias = config.attention_bias
        self.quant_config = quant_config
        self.qkv_proj = QKVParallelLinear(self.hidden_size, self.head_dim, self.total_num_heads,
                                          self.total_num_kv_heads, bias=self.qkv_bias)
        self.o_proj = RowParallelLinear(self.total_num_heads * self.head_dim, self.hidden_size, bias=False)
        self.use_qk_norm = getattr(config, "use_qk_norm", False)
        if self.use_qk_norm:
            self.q_norm = RMSNorm(self.head_dim, eps=config.rms_norm_eps)
            self.k_norm = RMSNorm(self.head_dim, eps=config.rms_norm_eps)
        partial_rotary_factor = getattr(config, "partial_rotary_factor", 0.5)
        rope_scaling = getattr(config, "rope_scaling", None)
        self.rotary_emb = get_rope(self.head_dim, rotary_dim=self.head_dim,
                                   max_position=config.max_position_embeddings,
                                   base=config.rope_theta, rope_scaling=rope_scaling,
                                   partial_rotary_factor=partial_rotary_factor)
        self.attn = Attention(self.num_heads, self.head_dim, self.scaling, self.num_kv_heads,
                              cache_config=None, quant_config=quant_config, prefix=f"{prefix}.attn")

    fuse_decode_attention = True

    @property
    def defer_qkv_reduce(self) -> bool:  # A/B switch (hip_ops option "defer_qkv_reduce")
        return hip_ops.opt("defer_qkv_reduce") != 0

    def _fusable(self) -> bool:
        lph = self.head_dim // 8
        return (self.head_dim % 8 == 0 and lph >= 2 and lph <= 64 and (lph & (lph - 1)) == 0
                and self.rotary_emb.rotary_dim % 16 == 0)

    def forward(self, hidden_states: torch.Tensor | None, positions: torch.Tensor, defer_o_reduce: bool = False,
                qkv: torch.Tensor | None = None, skip_o_proj: bool = False):
        """defer_o_reduce: the caller feeds the result straight into RMSNorm.forward(x, residual) and accepts a
        hip_ops.PartialSum (K-slice partial sums of o_proj, reduced inside that add-RMSNorm kernel).
        qkv: the QKV projection of this layer's normalised input, already computed by the caller (the one-token launch that
        fuses combine + input norm + projection); hidden_states is then unused (may be None).
        skip_o_proj: return the attention output [T, nh*hd] itself — the caller runs o_proj fused with what follows it
        (one token: o_proj + post-attention norm + router in one launch; TP = 1 only)."""
        ctx = get_context()
        attn = self.attn
        has_cache = attn.k_cache.numel() > 0 and attn.v_cache.numel() > 0
        two_d = qkv.dim() == 2 if qkv is not None else hidden_states.dim() == 2
        if (self.fuse_decode_attention and not ctx.is_prefill and has_cache and not self.use_qk_norm
                and self.head_dim == 128 and self.rotary_emb.rotary_dim == 64 and two_d
                and self.num_heads // self.num_kv_heads <= 16 and attn.k_cache.shape[1] % 32 == 0):
            # decode: rotary, K/V cache store and paged attention in one launch, straight from the QKV rows — or, at
            # decode batch sizes, from the QKV projection's K-slice partial sums (reduced + biased on load)
            bias = None
            if qkv is not None:
                pass
            elif self.defer_qkv_reduce and self.qkv_proj.can_defer_reduce(hidden_states):
                qkv = self.qkv_proj.forward_partial(hidden_states)
                bias = self.qkv_proj.bias.data if self.qkv_proj.bias is not None else None
            else:
                qkv = self.qkv_proj(hidden_states)
            o = hip_ops.paged_attn_decode_qkv(
                qkv, self.num_heads, self.num_kv_heads, self.head_dim, positions, self.rotary_emb.cos_sin_cache,
                self.rotary_emb.rotary_dim, ctx.slot_mapping, attn.k_cache, attn.v_cache, ctx.context_lens,
                ctx.block_tables, attn.scale, qkv_bias=bias)
            if skip_o_proj:
                return o.reshape(o.shape[0], -1)
            return self.o_proj(o.reshape(o.shape[0], -1), defer_reduce=defer_o_reduce)
        if (qkv is None and ctx.is_prefill and not self.use_qk_norm and self._fusable() and hidden_states.dim() == 2
                and self.qkv_proj.tp_size >= 1):
            # prefill row counts: split + rotary + K/V cache store ride in the QKV GEMM's epilogue (bit-identical to the two
            # launches below; None = shape outside that launch)
            fused = hip_ops.gemm_qkv_rope_store(
                hidden_states, self.qkv_proj.weight.data, self.qkv_proj.bias.data if self.qkv_proj.bias is not None else None,
                self.num_heads, self.num_kv_heads, self.head_dim, positions, self.rotary_emb.cos_sin_cache,
                self.rotary_emb.rotary_dim, attn.k_cache if has_cache else None, attn.v_cache if has_cache else None,
                ctx.slot_mapping if has_cache else None, ctx.block_tables is None)
            if fused is not None:
                o = attn.attend(*fused)
                if skip_o_proj:
                    return o.reshape(o.shape[0], -1)
                return self.o_proj(o.reshape(o.shape[0], -1), defer_reduce=defer_o_reduce)
        if qkv is None:
            qkv = self.qkv_proj(hidden_states)
        if self._fusable():
            need_kv = ctx.is_prefill and ctx.block_tables is None
            q, k, v = hip_ops.qkv_split_norm_rope_store(
                qkv, self.num_heads, self.num_kv_heads, self.head_dim, positions,
                self.rotary_emb.cos_sin_cache, self.rotary_emb.rotary_dim,
                self.q_norm.weight.data if self.use_qk_norm else None,
                self.k_norm.weight.data if self.use_qk_norm else None,
                float(self.config.rms_norm_eps),
                attn.k_cache if has_cache else None, attn.v_cache if has_cache else None,
                ctx.slot_mapping if has_cache else None, need_kv)
            o = attn.attend(q, k, v)
        else:  # unfused sequence of the same operators (odd head sizes)
            q, k, v = qkv.split([self.q_size, self.kv_size, self.kv_size], dim=-1)
            q = q.view(-1, self.num_heads, self.head_dim)
            k = k.view(-1, self.num_kv_heads, self.head_dim)
            v = v.view(-1, self.num_kv_heads, self.head_dim)
            if self.use_qk_norm:
                q, k = self.q_norm(q), self.k_norm(k)
            q, k = self.rotary_emb(positions, q, k)
            o = attn(q, k, v)
        if skip_o_proj:
            return o.reshape(o.shape[0], -1)
        return self.o_proj(o.reshape(o.shape[0], -1), defer_reduce=defer_o_reduce)

    def load_weights(self, path: str, prefix: str):
        sd = load_prefixed(path, prefix)
        if not sd:
            raise ValueError(f"No tensors with prefix '{prefix}' found in '{path}'")
        for name, shard in (("q_proj", "q"), ("k_proj", "k"), ("v_proj", "v")):
            if f"{prefix}.{name}.weight" in sd:
                self.qkv_proj.weight_loader(self.qkv_proj.weight, sd[f"{prefix}.{name}.weight"], shard)
            if self.config.attention_bias and f"{prefix}.{name}.bias" in sd:
                self.qkv_proj.weight_loader(self.qkv_proj.bias, sd[f"{prefix}.{name}.bias"], shard)
        if f"{prefix}.o_proj.weight" in sd:
            self.o_proj.weight_loader(self.o_proj.weight, sd[f"{prefix}.o_proj.weight"])
        if self.use_qk_norm:
            if f"{prefix}.q_norm.weight" in sd:
                self.q_norm.weight.data.copy_(sd[f"{prefix}.q_norm.weight"])
            if f"{prefix}.k_norm.weight" in sd:
                self.k_norm.weight.data.copy_(sd[f"{prefix}.k_norm.weight"])

    def load_kv_cache(self, kv_cache: dict):
        self.attn.k_cache = kv_cache["k_cache"]
        self.attn.v_cache = kv_cache["v_cache"]
