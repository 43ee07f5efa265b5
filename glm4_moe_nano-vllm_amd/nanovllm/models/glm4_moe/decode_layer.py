"""One GLM-4-MoE decoder layer = THE hot path — reference: models/glm4_moe/decode_layer.py:10-100.
MoE block iff layer_idx >= first_k_dense_replace, dense SiLU MLP otherwise; the MLP output is
returned un-added (the next layer's fused add-RMSNorm consumes it)."""
from __future__ import annotations

import torch
import torch.nn as nn

from nanovllm import hip_ops
from nanovllm.layers.layernorm import RMSNorm
from nanovllm.utils.loader import load_one
from .attention_new import Glm4MoeAttention
from .mlp import Glm4MoeMLP
from .moe import Glm4MoeMoE


class Glm4MoeDecoderLayer(nn.Module):

    def __init__(self, config, prefix: str = "") -> None:
        super().__init__()
        self.self_attn = Glm4MoeAttention(config, prefix=f"{prefix}.self_attn")
        try:
            layer_idx = int(prefix.split(sep=".")[-1])
        except ValueError:
            layer_idx = 0
        self.layer_idx = layer_idx
        if config.n_routed_experts is not None and layer_idx >= config.first_k_dense_replace:
            self.mlp = Glm4MoeMoE(config)
        else:
            self.mlp = Glm4MoeMLP(hidden_size=config.hidden_size,
                                  intermediate_size=config.intermediate_size,
                                  hidden_act=config.hidden_act, prefix=f"{prefix}.mlp")
        self.input_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)
        self.post_attention_layernorm = RMSNorm(config.hidden_size, eps=config.rms_norm_eps)

    def load_weights(self, model_path: str, prefix: str):
        for name, mod in (("input_layernorm", self.input_layernorm),
                          ("post_attention_layernorm", self.post_attention_layernorm)):
            w = load_one(model_path, f"{prefix}.{name}.weight")
            if w is not None:
                mod.weight.data.copy_(w)
        self.self_attn.load_weights(model_path, f"{prefix}.self_attn")
        self.mlp.load_weights(model_path, f"{prefix}.mlp")

    fuse_norm_router = True

    def forward(self, positions: torch.Tensor, hidden_states, residual: torch.Tensor | None,
                defer_combine: bool = False):
        """defer_combine: the caller passes the returned hidden state straight to the next layer (or to an
        RMSNorm.forward(x, residual)) and accepts a hip_ops.DeferredCombine instead of a tensor."""
        qkv = None
        if residual is not None and hip_ops.can_combine_norm_gemm(hidden_states, residual.shape[-1]):
            # one token (single-sequence decode): the previous layer's MoE combine, this layer's input add-RMSNorm and the
            # QKV projection in ONE launch — every workgroup of the GEMV recomputes the tiny combine + norm behind its
            # first weight loads; bit-identical to the separate launches
            ln, qp = self.input_layernorm, self.self_attn.qkv_proj
            qkv, residual = hip_ops.combine_add_rmsnorm_gemm(hidden_states, residual, ln.weight.data, ln.eps, qp.weight.data,
                                                             qp.bias.data if qp.bias is not None else None)
            hidden_states = None
        elif residual is None:
            hidden_states, residual = self.input_layernorm(hidden_states), hidden_states
        else:
            hidden_states, residual = self.input_layernorm(hidden_states, residual)
        gate = getattr(self.mlp, "gate", None)
        T1 = residual.reshape(-1, residual.shape[-1])
        op = self.self_attn.o_proj
        if (self.fuse_norm_router and gate is not None and T1.shape[0] == 1 and op.tp_size == 1 and op.bias is None
                and getattr(self.mlp, "ep_size", 1) == 1
                and hip_ops.can_gemm_router_tail(T1, residual.shape[-1], gate.weight.shape[0])):
            # ONE token: o_proj, post-attention add-RMSNorm, router and bucketing in one launch — the workgroup of the
            # o_proj GEMV that finishes last carries on with the token's norm + router (bit-identical to the two launches)
            o = self.self_attn(hidden_states, positions, qkv=qkv, skip_o_proj=True)
            ln = self.post_attention_layernorm
            fused = None
            if (o.dim() == 2 and o.shape[0] == 1 and o.is_contiguous()
                    and getattr(self.mlp, "expert_weight_dtype", None) is None
                    and hip_ops.can_decode_mlp_engine(o, residual.shape[-1], gate.weight.shape[0], gate.top_k,
                                                      self.mlp.config.moe_intermediate_size)):
                # r05: ... and the grouped gate|up and down projections too — the whole MLP half of the layer as ONE
                # persistent launch (csrc/decode_engine.hip), bit-identical to the three launches below
                w13, w2 = self.mlp._stacked()
                eng = hip_ops.decode_mlp_engine(o, op.weight.data, residual, ln.weight.data, ln.eps, gate.weight.data,
                                                gate.e_score_correction_bias, gate.top_k, float(gate.routed_scaling_factor),
                                                w13, w2)
                if eng is not None:
                    hidden_states, residual, idx, w, aligned, y, _ = eng
                    # the block's module call stays (hooks, same return types): the expert rows are already there
                    hidden_states = self.mlp(hidden_states, routing=(idx, w), aligned=aligned, defer_combine=defer_combine,
                                             expert_rows=y)
                    return hidden_states, residual
            if o.dim() == 2 and o.shape[0] == 1 and o.is_contiguous():
                fused = hip_ops.gemm_add_rmsnorm_router_topk_align(
                    o, op.weight.data, residual, ln.weight.data, ln.eps, gate.weight.data, gate.e_score_correction_bias,
                    gate.top_k, float(gate.routed_scaling_factor), True)
            if fused is not None:
                hidden_states, residual, idx, w, aligned = fused
                hidden_states = self.mlp(hidden_states, routing=(idx, w), aligned=aligned, defer_combine=defer_combine)
                return hidden_states, residual
            hidden_states = op(o, defer_reduce=True)  # shape outside the fused launch's window: two launches
        else:
            # o_proj may hand over K-slice partial sums; the add-RMSNorm right below reduces them (decode batches)
            hidden_states = self.self_attn(hidden_states, positions, defer_o_reduce=True, qkv=qkv)
        if self.fuse_norm_router and gate is not None and residual.shape[-1] <= 8192:
            # post-attention add-RMSNorm and the router of the same token in one launch
            ln = self.post_attention_layernorm
            # ... and the token -> expert bucketing (moe_align), done by that launch's last workgroup
            # (not under expert parallelism: every rank buckets its own token slice, distributed/expert_parallel.py)
            want_align = True if getattr(self.mlp, "ep_size", 1) == 1 else None
            hidden_states, residual, idx, w, *aligned = hip_ops.add_rmsnorm_router_topk(
                hidden_states, residual, ln.weight.data, ln.eps, gate.weight.data, gate.e_score_correction_bias,
                gate.top_k, float(gate.routed_scaling_factor), align_with_shared=want_align)
            hidden_states = self.mlp(hidden_states, routing=(idx, w), aligned=aligned[0] if aligned else None,
                                     defer_combine=defer_combine)
        else:
            hidden_states, residual = self.post_attention_layernorm(hidden_states, residual)
            if gate is not None:
                hidden_states = self.mlp(hidden_states, defer_combine=defer_combine)
            else:
                hidden_states = self.mlp(hidden_states)
        return hidden_states, residual
