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
