"""TEST INFRASTRUCTURE ONLY (same rule as glm4_moe_oracle.py: imported by tests/, smoke() and bench.py's
cpu_baseline leg, never by the product).  Copies the tensors of the product's GPU modules into the oracle's
plain containers, so a parity test and the CPU baseline evaluate the reference algorithm on exactly the bytes the
HIP path ran on."""
from __future__ import annotations

import torch

from . import glm4_moe_oracle as O

CFG_KEYS = ("num_attention_heads", "num_key_value_heads", "head_dim", "rms_norm_eps", "num_experts_per_tok",
            "routed_scaling_factor", "partial_rotary_factor", "n_routed_experts")


def cfg_dict(cfg) -> dict:
    """The config fields the oracle reads, from a namespace / HF config / dict."""
    get = cfg.get if isinstance(cfg, dict) else (lambda k, d=None: getattr(cfg, k, d))
    return {k: get(k) for k in CFG_KEYS if get(k) is not None}


def _cpu(t):
    return None if t is None else t.detach().to("cpu").contiguous()


def layer_weights_from_module(layer, n_routed_experts: int | None = None) -> O.LayerWeights:
    """nanovllm.models.glm4_moe.decode_layer.Glm4MoeDecoderLayer (on the GPU) -> O.LayerWeights (CPU)."""
    att = layer.self_attn
    kw = dict(input_ln=_cpu(layer.input_layernorm.weight), post_ln=_cpu(layer.post_attention_layernorm.weight),
              qkv_w=_cpu(att.qkv_proj.weight), qkv_b=_cpu(att.qkv_proj.bias), o_w=_cpu(att.o_proj.weight),
              q_norm=_cpu(att.q_norm.weight) if getattr(att, "use_qk_norm", False) else None,
              k_norm=_cpu(att.k_norm.weight) if getattr(att, "use_qk_norm", False) else None)
    mlp = layer.mlp
    if hasattr(mlp, "gate"):
        fp8 = getattr(mlp, "expert_weight_dtype", None) == "fp8_e4m3"
        if fp8 and mlp._fp8 is None:
            mlp.quantize_experts_fp8()
        E = n_routed_experts or len(mlp.experts)
        kw.update(gate_w=_cpu(mlp.gate.weight), gate_bias=_cpu(mlp.gate.e_score_correction_bias))
        if not getattr(mlp, "_bf16_freed", False):  # fp8 experts release their bf16 stacks; the oracle then uses `fp8` only
            w13, w2 = mlp._stacked()
            kw.update(w13=_cpu(w13[:E]), w2=_cpu(w2[:E]), shared_w13=_cpu(w13[E]), shared_w2=_cpu(w2[E]))
        if getattr(mlp, "expert_weight_dtype", None) == "fp8_e4m3":
            q13, s13, q2, s2 = mlp._fp8  # uint8 views with a padded pitch -> contiguous float8 on the CPU
            f8 = lambda q: q.cpu().contiguous().view(torch.float8_e4m3fn)
            kw["fp8"] = (f8(q13), s13.cpu(), f8(q2), s2.cpu())
    else:
        kw.update(dense_w13=_cpu(mlp.gate_up_proj.weight), dense_w2=_cpu(mlp.down_proj.weight))
    return O.LayerWeights(**kw)


def cos_sin_cache(cfg) -> torch.Tensor:
    c = cfg if isinstance(cfg, dict) else vars(cfg)
    rd = O.rotary_dim_of(c["head_dim"], c.get("partial_rotary_factor", 0.5))
    return O.build_cos_sin_cache(rd, c["max_position_embeddings"], c["rope_theta"])
