"""Shared helpers for the test-suite: golden fixtures and oracle weight containers."""
from __future__ import annotations

import json
import os

import torch
from safetensors import safe_open

from oracle import glm4_moe_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name: str):
    path = os.path.join(GOLDEN, name + ".safetensors")
    out = {}
    with safe_open(path, framework="pt", device="cpu") as f:
        meta = json.loads(f.metadata()["meta"])
        for k in f.keys():
            out[k] = f.get_tensor(k)
    return out, meta


def layer_weights_from_golden(t: dict, cfg: dict, prefix: str = "layer.") -> O.LayerWeights:
    g = lambda k: t.get(prefix + k)
    kw = dict(input_ln=g("input_layernorm.weight"), post_ln=g("post_attention_layernorm.weight"),
              qkv_w=g("self_attn.qkv_proj.weight"), qkv_b=g("self_attn.qkv_proj.bias"),
              o_w=g("self_attn.o_proj.weight"), q_norm=g("self_attn.q_norm.weight"),
              k_norm=g("self_attn.k_norm.weight"))
    if g("mlp.gate.weight") is not None:
        E = cfg["n_routed_experts"]
        kw.update(gate_w=g("mlp.gate.weight"), gate_bias=g("mlp.gate.e_score_correction_bias"),
                  w13=torch.stack([g(f"mlp.experts.{e}.gate_up_proj.weight") for e in range(E)]),
                  w2=torch.stack([g(f"mlp.experts.{e}.down_proj.weight") for e in range(E)]),
                  shared_w13=g("mlp.shared_experts.gate_up_proj.weight"),
                  shared_w2=g("mlp.shared_experts.down_proj.weight"))
    else:
        kw.update(dense_w13=g("mlp.gate_up_proj.weight"), dense_w2=g("mlp.down_proj.weight"))
    return O.LayerWeights(**kw)


def bf16_ulp_diff(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Distance in bf16 units-in-the-last-place between two bf16 tensors (sign-magnitude order)."""
    def key(t):
        i = t.contiguous().view(torch.int16).to(torch.int32) & 0xFFFF
        return torch.where(i >= 0x8000, 0x8000 - i, i)
    return (key(a) - key(b)).abs()


def assert_close_bf16(actual, expected, max_ulp=1, frac_inexact=0.02, atol=0.0, what=""):
    """Elementwise |ulp distance| <= max_ulp (or |diff| <= atol), and at most `frac_inexact` of
    the elements differ at all."""
    a, e = actual.detach().cpu(), expected.detach().cpu()
    assert a.shape == e.shape, f"{what}: shape {a.shape} vs {e.shape}"
    # a NaN compares False in the atol clause below and would pass unseen
    assert bool(torch.isfinite(e.float()).all()) and bool(torch.isfinite(a.float()).all()), f"{what}: non-finite elements"
    d = bf16_ulp_diff(a, e)
    bad = (d > max_ulp) & ((a.float() - e.float()).abs() > atol)
    assert not bool(bad.any()), (
        f"{what}: {int(bad.sum())} elements differ by > {max_ulp} bf16 ulp "
        f"(max {int(d.max())} ulp, max abs {float((a.float() - e.float()).abs().max()):.4g})")
    frac = float((d > 0).float().mean())
    assert frac <= frac_inexact, f"{what}: {frac:.4%} of elements inexact (allowed {frac_inexact:.2%})"


# ---------------------------------------------------------------------------------------------
# product-side helpers (GPU): build the nanovllm modules from a golden fixture
def build_product_layer(t: dict, cfg: dict, prefix: str, device="cuda"):
    """Instantiate nanovllm.models.glm4_moe.decode_layer.Glm4MoeDecoderLayer the way the
    reference's runner does (default dtype bf16, default device cuda; model_runner.py:30-31)
    and load the fixture's weights through the ordinary state-dict surface."""
    import types
    from nanovllm.layers.rotary_embedding import get_rope
    from nanovllm.models.glm4_moe.decode_layer import Glm4MoeDecoderLayer
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    get_rope.cache_clear()
    try:
        with torch.device(device):
            layer = Glm4MoeDecoderLayer(types.SimpleNamespace(**cfg), prefix=prefix)
    finally:
        torch.set_default_dtype(old)
    sd = {k[len("layer."):]: v for k, v in t.items() if k.startswith("layer.")}
    missing, unexpected = layer.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("cos_sin" in m for m in missing), missing
    return layer


def router_margin(choice: torch.Tensor, k: int) -> torch.Tensor:
    """Per row: k-th largest minus (k+1)-th largest routing score (fp32)."""
    s = torch.sort(choice.float(), dim=-1, descending=True)[0]
    return s[:, k - 1] - s[:, k]


# ---------------------------------------------------------------------------------------------
# tiny synthetic GLM-4-MoE checkpoint (HF key names) for the engine tests
def tiny_model_config(**over) -> dict:
    cfg = dict(model_type="glm4_moe", hidden_size=128, num_attention_heads=4, num_key_value_heads=2,
               head_dim=128, attention_bias=True, use_qk_norm=False, partial_rotary_factor=0.5,
               rope_theta=10000.0, rope_scaling=None, max_position_embeddings=1024, rms_norm_eps=1e-5,
               n_routed_experts=8, num_experts_per_tok=2, n_group=1, topk_group=1,
               routed_scaling_factor=1.0, moe_intermediate_size=64, intermediate_size=192,
               first_k_dense_replace=1, hidden_act="silu", vocab_size=512, num_hidden_layers=2,
               tie_word_embeddings=False, torch_dtype="bfloat16")
    cfg.update(over)
    return cfg


def write_tiny_checkpoint(path: str, cfg: dict, seed: int = 0) -> dict:
    """Writes config.json + model.safetensors; returns the tensor dict (CPU, bf16)."""
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    g = torch.Generator().manual_seed(seed)
    H, nh, nkv, hd = cfg["hidden_size"], cfg["num_attention_heads"], cfg["num_key_value_heads"], cfg["head_dim"]
    I, Id, E, V = cfg["moe_intermediate_size"], cfg["intermediate_size"], cfg["n_routed_experts"], cfg["vocab_size"]
    bf = torch.bfloat16
    rn = lambda *s, std=0.08: (torch.randn(*s, generator=g) * std).to(bf)
    ln = lambda n: (1 + 0.05 * torch.randn(n, generator=g)).to(bf)
    t = {"model.embed_tokens.weight": rn(V, H, std=1.0), "model.norm.weight": ln(H),
         "lm_head.weight": rn(V, H, std=0.1)}
    for i in range(cfg["num_hidden_layers"]):
        p = f"model.layers.{i}"
        t[f"{p}.input_layernorm.weight"], t[f"{p}.post_attention_layernorm.weight"] = ln(H), ln(H)
        for name, rows in (("q_proj", nh * hd), ("k_proj", nkv * hd), ("v_proj", nkv * hd)):
            t[f"{p}.self_attn.{name}.weight"] = rn(rows, H)
            t[f"{p}.self_attn.{name}.bias"] = rn(rows, std=0.1)
        t[f"{p}.self_attn.o_proj.weight"] = rn(H, nh * hd)
        if cfg.get("use_qk_norm"):
            t[f"{p}.self_attn.q_norm.weight"], t[f"{p}.self_attn.k_norm.weight"] = ln(hd), ln(hd)
        if i < cfg["first_k_dense_replace"]:
            t[f"{p}.mlp.gate_proj.weight"], t[f"{p}.mlp.up_proj.weight"] = rn(Id, H), rn(Id, H)
            t[f"{p}.mlp.down_proj.weight"] = rn(H, Id)
        else:
            t[f"{p}.mlp.gate.weight"] = rn(E, H, std=H ** -0.5)
            t[f"{p}.mlp.gate.e_score_correction_bias"] = (torch.rand(E, generator=g) * 0.2 - 0.1).float()
            for e in list(range(E)) + ["shared"]:
                q = f"{p}.mlp.shared_experts" if e == "shared" else f"{p}.mlp.experts.{e}"
                t[f"{q}.gate_proj.weight"], t[f"{q}.up_proj.weight"] = rn(I, H), rn(I, H)
                t[f"{q}.down_proj.weight"] = rn(H, I)
    save_file({k: v.contiguous() for k, v in t.items()}, os.path.join(path, "model.safetensors"))
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cfg, f)
    return t


def oracle_layers_from_checkpoint(t: dict, cfg: dict) -> list:
    layers = []
    E = cfg["n_routed_experts"]
    for i in range(cfg["num_hidden_layers"]):
        p = f"model.layers.{i}"
        cat = lambda *names: torch.cat([t[n] for n in names])
        kw = dict(input_ln=t[f"{p}.input_layernorm.weight"], post_ln=t[f"{p}.post_attention_layernorm.weight"],
                  qkv_w=cat(f"{p}.self_attn.q_proj.weight", f"{p}.self_attn.k_proj.weight", f"{p}.self_attn.v_proj.weight"),
                  qkv_b=cat(f"{p}.self_attn.q_proj.bias", f"{p}.self_attn.k_proj.bias", f"{p}.self_attn.v_proj.bias"),
                  o_w=t[f"{p}.self_attn.o_proj.weight"], q_norm=t.get(f"{p}.self_attn.q_norm.weight"),
                  k_norm=t.get(f"{p}.self_attn.k_norm.weight"))
        if i < cfg["first_k_dense_replace"]:
            kw.update(dense_w13=cat(f"{p}.mlp.gate_proj.weight", f"{p}.mlp.up_proj.weight"), dense_w2=t[f"{p}.mlp.down_proj.weight"])
        else:
            ex = lambda e: f"{p}.mlp.experts.{e}"
            kw.update(gate_w=t[f"{p}.mlp.gate.weight"], gate_bias=t[f"{p}.mlp.gate.e_score_correction_bias"],
                      w13=torch.stack([cat(f"{ex(e)}.gate_proj.weight", f"{ex(e)}.up_proj.weight") for e in range(E)]),
                      w2=torch.stack([t[f"{ex(e)}.down_proj.weight"] for e in range(E)]),
                      shared_w13=cat(f"{p}.mlp.shared_experts.gate_proj.weight", f"{p}.mlp.shared_experts.up_proj.weight"),
                      shared_w2=t[f"{p}.mlp.shared_experts.down_proj.weight"])
        layers.append(O.LayerWeights(**kw))
    return layers


def oracle_next_token_logits(t: dict, cfg: dict, layers, token_ids: list[int]) -> torch.Tensor:
    """Full-sequence (cache-less prefill) forward on the CPU oracle; logits of the last position."""
    ids = torch.tensor(token_ids, dtype=torch.int64)
    pos = torch.arange(len(token_ids), dtype=torch.int64)
    cu = torch.tensor([0, len(token_ids)], dtype=torch.int32)
    cs = O.build_cos_sin_cache(O.rotary_dim_of(cfg["head_dim"], cfg["partial_rotary_factor"]),
                               cfg["max_position_embeddings"], cfg["rope_theta"])
    meta = O.AttnMeta(True, cu_seqlens_q=cu, cu_seqlens_k=cu)
    h = O.model_forward(ids, pos, t["model.embed_tokens.weight"], layers, t["model.norm.weight"], cfg, cs, None, meta)
    return O.linear(h[-1:], t["lm_head.weight"])[0]
