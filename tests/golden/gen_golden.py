"""Mint golden vectors by running the REFERENCE implementation (read-only at /root/reference)
on CPU in the build container.  The reference never travels: only the small tensors written
here (inputs, weights, expected outputs) are committed, together with this script.

    python tests/golden/gen_golden.py            # rewrites tests/golden/*.safetensors

Harness-side shims (no reference file is edited; recipe recorded in SURVEY.md §8c):
  1. a stub ``flash_attn`` module (the wheel is not installed; arithmetic of that third-party
     dependency is PARITY UNPINNED) implementing fp32 softmax(QK^T)V per sequence;
  2. ``safetensors.torch.load_file`` answers the debug reads of /data/ai_infra/... that
     moe.py:153-157 and attention_new.py:288-292 perform on every forward with dummies;
  3. configs are ``types.SimpleNamespace`` (transformers 5.x Glm4MoeConfig lacks rope_theta);
  4. ``store_kvcache`` (a Triton kernel) is replaced by an index copy for CPU;
  5. ``torch.set_default_dtype(torch.bfloat16)`` reproduces the runner's dtype regime
     (engine/model_runner.py:30).
"""
from __future__ import annotations

import json
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import glm4_moe_oracle as O  # noqa: E402  (stand-in attention math for the stub)


def _install_shims():
    fa = types.ModuleType("flash_attn")
    fa.__spec__ = types.SimpleNamespace(name="flash_attn", loader=None, origin="stub",
                                        submodule_search_locations=None, parent="", cached=None,
                                        has_location=False)

    def flash_attn_varlen_func(q, k, v, max_seqlen_q=None, cu_seqlens_q=None, max_seqlen_k=None,
                               cu_seqlens_k=None, softmax_scale=None, causal=True,
                               block_table=None):
        assert causal
        return O.attn_prefill_varlen(q, k, v, cu_seqlens_q, cu_seqlens_k, softmax_scale,
                                     block_table)

    def flash_attn_with_kvcache(q, k_cache, v_cache, cache_seqlens=None, block_table=None,
                                softmax_scale=None, causal=True):
        o = O.attn_decode_paged(q[:, 0], k_cache, v_cache, cache_seqlens, block_table,
                                softmax_scale)
        return o.unsqueeze(1)

    fa.flash_attn_varlen_func = flash_attn_varlen_func
    fa.flash_attn_with_kvcache = flash_attn_with_kvcache
    sys.modules["flash_attn"] = fa

    import safetensors.torch as st
    real_load = st.load_file

    class _Dummy(dict):
        def __missing__(self, key):
            return torch.zeros(1)

    def load_file(path, *a, **kw):
        if str(path).startswith("/data/ai_infra"):
            return _Dummy()
        return real_load(path, *a, **kw)

    st.load_file = load_file
    import safetensors
    safetensors.torch.load_file = load_file


def _cfg(**over):
    base = dict(hidden_size=128, num_attention_heads=4, num_key_value_heads=2, head_dim=128,
                attention_bias=True, use_qk_norm=False, partial_rotary_factor=0.5,
                rope_theta=10000.0, rope_scaling=None, max_position_embeddings=1024,
                rms_norm_eps=1e-5, n_routed_experts=8, num_experts_per_tok=2, n_group=1,
                topk_group=1, routed_scaling_factor=1.0, moe_intermediate_size=64,
                intermediate_size=192, first_k_dense_replace=1, hidden_act="silu",
                norm_topk_prob=True)
    base.update(over)
    return types.SimpleNamespace(**base)


def _init(module, gen, std=0.08):
    for name, p in module.named_parameters():
        if "layernorm" in name or name.endswith("norm.weight"):
            p.data.copy_((1.0 + 0.05 * torch.randn(p.shape, generator=gen)).to(p.dtype))
        elif name.endswith("bias"):
            p.data.copy_((0.1 * torch.randn(p.shape, generator=gen)).to(p.dtype))
        else:
            p.data.copy_((std * torch.randn(p.shape, generator=gen)).to(p.dtype))
    for name, b in module.named_buffers():
        if name.endswith("e_score_correction_bias"):
            b.copy_((torch.rand(b.shape, generator=gen) * 0.2 - 0.1).to(b.dtype))


def _save(name, tensors, meta):
    from safetensors.torch import save_file
    out = {k: v.detach().clone().contiguous() for k, v in tensors.items()}
    save_file(out, os.path.join(HERE, name + ".safetensors"),
              metadata={"meta": json.dumps(meta)})
    nbytes = sum(v.numel() * v.element_size() for v in out.values())
    print(f"  {name}: {len(out)} tensors, {nbytes / 1024:.0f} KiB")


def _layer_tensors(layer, prefix=""):
    return {prefix + k: v for k, v in layer.state_dict().items()} | {
        prefix + k: v for k, v in layer.named_buffers() if k.endswith("e_score_correction_bias")}


def main():
    _install_shims()
    sys.path.insert(0, REF)
    torch.set_default_dtype(torch.bfloat16)
    torch.manual_seed(0)
    gen = torch.Generator().manual_seed(1234)

    from nanovllm.layers.layernorm import RMSNorm as LNRMSNorm
    from nanovllm.layers.RMSNorm import RMSNorm as HeadRMSNorm
    from nanovllm.layers.rotary_embedding import RotaryEmbedding
    from nanovllm.layers.activation import SiluAndMul
    from nanovllm.layers.linear import ReplicatedLinear
    import nanovllm.layers.attention as ref_attention
    from nanovllm.models.glm4_moe.moe import GlmMoeSelectTopk, Glm4MoeMoE
    from nanovllm.models.glm4_moe.mlp import Glm4MoeMLP
    from nanovllm.models.glm4_moe.decode_layer import Glm4MoeDecoderLayer
    from nanovllm.utils.context import set_context, reset_context

    ref_attention.store_kvcache = O.store_kvcache  # shim 4 (Triton kernel cannot run on CPU)
    bf = torch.bfloat16

    with torch.inference_mode():
        # ---------------- elementwise ops ----------------
        T, H = 5, 128
        x = torch.randn(T, H, generator=gen).to(bf)
        res = torch.randn(T, H, generator=gen).to(bf)
        ln = LNRMSNorm(H, eps=1e-5)
        _init(ln, gen)
        y_rms = ln(x.clone())
        y_add, r_add = ln(x.clone(), res.clone())
        hn = HeadRMSNorm(128, eps=1e-5)
        hn.weight.data.copy_((1.0 + 0.05 * torch.randn(128, generator=gen)).to(bf))
        xh = torch.randn(T, 4, 128, generator=gen).to(bf)
        y_head = hn(xh)
        rope = RotaryEmbedding(128, 64, 1024, 10000.0)
        pos = torch.tensor([0, 1, 17, 511, 1023], dtype=torch.int64)
        q = torch.randn(T, 4, 128, generator=gen).to(bf)
        k = torch.randn(T, 2, 128, generator=gen).to(bf)
        q_r, k_r = rope(pos, q, k)
        rope_full = RotaryEmbedding(16, 16, 64, 10000.0)  # C1-like tiny head, full rotary
        pos2 = torch.tensor([3, 0, 63], dtype=torch.int64)
        q2 = torch.randn(3, 4, 16, generator=gen).to(bf)
        k2 = torch.randn(3, 2, 16, generator=gen).to(bf)
        q2_r, k2_r = rope_full(pos2, q2, k2)
        xs = (3.0 * torch.randn(T, 2 * 96, generator=gen)).to(bf)
        y_silu = SiluAndMul()(xs)
        lin = ReplicatedLinear(H, 72, bias=True)
        _init(lin, gen)
        y_lin = lin(x)
        _save("ops", dict(x=x, res=res, ln_w=ln.weight, y_rms=y_rms, y_add=y_add, r_add=r_add,
                          hn_w=hn.weight, xh=xh, y_head=y_head, pos=pos, q=q, k=k, q_r=q_r,
                          k_r=k_r, cos_sin=rope.cos_sin_cache, pos2=pos2, q2=q2, k2=k2,
                          q2_r=q2_r, k2_r=k2_r, cos_sin2=rope_full.cos_sin_cache, xs=xs,
                          y_silu=y_silu, lin_w=lin.weight, lin_b=lin.bias, y_lin=y_lin),
              dict(eps=1e-5, rotary_dim=64, rope_base=10000.0))

        # ---------------- router ----------------
        tens, meta = {}, {}
        for tag, (E, kk, Hh, Tt, zero_bias, rsf) in {
                "e64k8": (64, 8, 256, 48, False, 1.0),
                "e8k2": (8, 2, 128, 33, False, 2.5),
                "e128k8": (128, 8, 128, 16, False, 1.0),
                "ties": (64, 8, 64, 16, True, 1.0)}.items():
            cfg = _cfg(hidden_size=Hh, n_routed_experts=E, num_experts_per_tok=kk,
                       routed_scaling_factor=rsf)
            gate = GlmMoeSelectTopk(cfg)
            _init(gate, gen, std=1.0 / Hh ** 0.5)
            if zero_bias:
                gate.e_score_correction_bias.zero_()
            xr = torch.randn(Tt, Hh, generator=gen).to(bf)
            idx, w = gate(xr)
            tens |= {f"{tag}.x": xr, f"{tag}.gate_w": gate.weight,
                     f"{tag}.bias": gate.e_score_correction_bias, f"{tag}.idx": idx, f"{tag}.w": w}
            meta[tag] = dict(E=E, k=kk, H=Hh, T=Tt, rsf=rsf)
        _save("router", tens, meta)

        # ---------------- expert MLP and MoE block ----------------
        m = Glm4MoeMLP(hidden_size=128, intermediate_size=64, hidden_act="silu")
        _init(m, gen)
        xm = torch.randn(7, 128, generator=gen).to(bf)
        _save("mlp", dict(x=xm, w13=m.gate_up_proj.weight, w2=m.down_proj.weight, y=m(xm)), {})

        cfg = _cfg()
        moe = Glm4MoeMoE(cfg)
        _init(moe, gen)
        xe = torch.randn(19, 128, generator=gen).to(bf)
        idx, w = moe.gate(xe)
        ye = moe(xe)
        _save("moe", _layer_tensors(moe) | dict(x=xe, y=ye, idx=idx, w=w), vars(cfg))

        # ---------------- decoder layer: prefill (2 sequences), MoE and dense variants -------
        for tag, prefix, over in (("layer_moe_prefill", "model.layers.1", {}),
                                  ("layer_dense_prefill", "model.layers.0", {}),
                                  ("layer_moe_prefill_qknorm", "model.layers.1",
                                   dict(use_qk_norm=True))):
            cfg = _cfg(**over)
            layer = Glm4MoeDecoderLayer(cfg, prefix=prefix)
            _init(layer, gen)
            lens = [21, 16]
            Tt = sum(lens)
            h = torch.randn(Tt, 128, generator=gen).to(bf)
            positions = torch.cat([torch.arange(n) for n in lens]).to(torch.int64)
            cu = torch.tensor([0, 21, 37], dtype=torch.int32)
            kc = torch.zeros(2, 256, 2, 128)
            vc = torch.zeros(2, 256, 2, 128)
            slots = torch.cat([256 * 1 + torch.arange(21), 256 * 0 + torch.arange(16)]).to(torch.int32)
            layer.self_attn.attn.k_cache, layer.self_attn.attn.v_cache = kc, vc
            set_context(True, cu, cu, 21, 21, slots, None, None)
            out1, res1 = layer(positions, h.clone(), None)
            resin = torch.randn(Tt, 128, generator=gen).to(bf)
            out2, res2 = layer(positions, h.clone(), resin.clone())
            reset_context()
            _save(tag, _layer_tensors(layer, "layer.") | dict(
                h=h, positions=positions, cu_seqlens=cu, slot_mapping=slots, out_first=out1,
                res_first=res1, res_in=resin, out=out2, res=res2,
                k_rows=kc.view(-1, 256)[slots.long()], v_rows=vc.view(-1, 256)[slots.long()],
                cos_sin=layer.self_attn.rotary_emb.cos_sin_cache), vars(cfg) | dict(lens=lens))

        # ---------------- decoder layer: decode over a paged cache (ragged lengths) ----------
        for tag, over in (("layer_moe_decode", {}), ("layer_moe_decode_qknorm", dict(use_qk_norm=True))):
            cfg = _cfg(**over)
            layer = Glm4MoeDecoderLayer(cfg, prefix="model.layers.1")
            _init(layer, gen)
            lens = [300, 256, 5]  # context_lens include the token written by this step
            B = len(lens)
            kc = torch.randn(4, 256, 2, 128, generator=gen).to(bf)
            vc = torch.randn(4, 256, 2, 128, generator=gen).to(bf)
            bt = torch.tensor([[2, 0], [3, -1], [1, -1]], dtype=torch.int32)
            slots = torch.tensor([0 * 256 + 43, 3 * 256 + 255, 1 * 256 + 4], dtype=torch.int32)
            positions = torch.tensor([n - 1 for n in lens], dtype=torch.int64)
            ctx = torch.tensor(lens, dtype=torch.int32)
            h = torch.randn(B, 128, generator=gen).to(bf)
            resin = torch.randn(B, 128, generator=gen).to(bf)
            kc_in, vc_in = kc.clone(), vc.clone()
            layer.self_attn.attn.k_cache, layer.self_attn.attn.v_cache = kc, vc
            set_context(False, slot_mapping=slots, context_lens=ctx, block_tables=bt)
            out, res = layer(positions, h.clone(), resin.clone())
            reset_context()
            _save(tag, _layer_tensors(layer, "layer.") | dict(
                h=h, res_in=resin, positions=positions, context_lens=ctx, block_tables=bt,
                slot_mapping=slots, k_cache_in=kc_in, v_cache_in=vc_in, out=out, res=res,
                k_rows=kc.view(-1, 256)[slots.long()], v_rows=vc.view(-1, 256)[slots.long()],
                cos_sin=layer.self_attn.rotary_emb.cos_sin_cache), vars(cfg) | dict(lens=lens))
        # ---------------- router: exact ties ON the k-th / (k+1)-th boundary (r03) ----------------
        # Own generator and own file: the nine fixtures above stay byte-identical.  Gate rows come in identical
        # triples with one correction bias per triple, so every score of the reference comes three times; with k = 8
        # the sorted scores are a a a b b b c c | c ...: on EVERY row the 8th and the 9th score are exactly equal and
        # torch.topk (moe.py:39) has to pick two of three equal candidates.  The experts left over (E mod 3) carry a
        # bias of -10 and are never candidates.  The triples are scattered over the expert ids by a fixed permutation.
        gen2 = torch.Generator().manual_seed(4321)
        tens, meta = {}, {}
        for tag, (E, kk, Hh, Tt, rsf) in {"e64k8": (64, 8, 128, 24, 1.0), "e128k8": (128, 8, 256, 24, 2.5)}.items():
            cfg = _cfg(hidden_size=Hh, n_routed_experts=E, num_experts_per_tok=kk, routed_scaling_factor=rsf)
            gate = GlmMoeSelectTopk(cfg)
            ng = E // 3
            base = (torch.randn(ng, Hh, generator=gen2) / Hh ** 0.5).to(bf)
            gbias = torch.rand(ng, generator=gen2) * 0.2 - 0.1
            perm = torch.randperm(E, generator=gen2)
            wfull = (torch.randn(E, Hh, generator=gen2) / Hh ** 0.5).to(bf)
            bfull = torch.full((E,), -10.0)
            group = torch.full((E,), -1, dtype=torch.int64)
            for j in range(3 * ng):
                wfull[perm[j]] = base[j // 3]
                bfull[perm[j]] = gbias[j // 3]
                group[perm[j]] = j // 3
            gate.weight.data.copy_(wfull)
            gate.e_score_correction_bias.copy_(bfull)
            xr = torch.randn(Tt, Hh, generator=gen2).to(bf)
            idx, w = gate(xr)
            tens |= {f"{tag}.x": xr, f"{tag}.gate_w": gate.weight, f"{tag}.bias": gate.e_score_correction_bias,
                     f"{tag}.idx": idx, f"{tag}.w": w, f"{tag}.group": group}
            meta[tag] = dict(E=E, k=kk, H=Hh, T=Tt, rsf=rsf)
        _save("router_ties", tens, meta)

        # ---------------- BASELINE config 1 (r04): ONE MoE decoder layer, 2 experts top-1, bs=1 seq=128 ----------------
        # The reference's own smoke of this path is examples/decode_layer.py:7-55 (first-layer form, residual=None).
        # Own generator and own file.  Three evaluations by the reference: (i) the 128-token prompt in the runner's
        # bf16 regime, K/V written to block 1 of a paged cache; (ii) the decode step that follows it (token 129 attends
        # the 128 cached rows + its own); (iii) the same prompt with the SAME weights in fp32 modules (the "CPU eager"
        # evaluation BASELINE.json names; SURVEY.md 8c probe (i)).
        _c1(gen=torch.Generator().manual_seed(1001), Glm4MoeDecoderLayer=Glm4MoeDecoderLayer,
            set_context=set_context, reset_context=reset_context)
    print("done")


def _c1(gen, Glm4MoeDecoderLayer, set_context, reset_context):
    bf = torch.bfloat16
    cfg = _cfg(n_routed_experts=2, num_experts_per_tok=1)
    layer = Glm4MoeDecoderLayer(cfg, prefix="model.layers.1")
    _init(layer, gen)
    N = 128
    h = torch.randn(N, 128, generator=gen).to(bf)
    positions = torch.arange(N, dtype=torch.int64)
    cu = torch.tensor([0, N], dtype=torch.int32)
    kc = torch.zeros(2, 256, 2, 128)
    vc = torch.zeros(2, 256, 2, 128)
    slots = (256 * 1 + torch.arange(N)).to(torch.int32)
    layer.self_attn.attn.k_cache, layer.self_attn.attn.v_cache = kc, vc
    set_context(True, cu, cu, N, N, slots, None, None)
    out, res = layer(positions, h.clone(), None)
    reset_context()
    k_rows, v_rows = kc.view(-1, 256)[slots.long()].clone(), vc.view(-1, 256)[slots.long()].clone()
    # (ii) decode step over the cache the prompt has just filled
    kc_in, vc_in = kc.clone(), vc.clone()
    h1 = torch.randn(1, 128, generator=gen).to(bf)
    r1 = torch.randn(1, 128, generator=gen).to(bf)
    pos1 = torch.tensor([N], dtype=torch.int64)
    ctx1 = torch.tensor([N + 1], dtype=torch.int32)
    bt1 = torch.tensor([[1]], dtype=torch.int32)
    slot1 = torch.tensor([256 + N], dtype=torch.int32)
    set_context(False, slot_mapping=slot1, context_lens=ctx1, block_tables=bt1)
    out_d, res_d = layer(pos1, h1.clone(), r1.clone())
    reset_context()
    # (iii) fp32 modules, same weight values
    torch.set_default_dtype(torch.float32)
    try:
        layer32 = Glm4MoeDecoderLayer(cfg, prefix="model.layers.1")
        layer32.load_state_dict(layer.state_dict())
        layer32.mlp.gate.e_score_correction_bias.copy_(layer.mlp.gate.e_score_correction_bias)
        assert layer32.self_attn.qkv_proj.weight.dtype == torch.float32
        cu32 = cu.clone()
        set_context(True, cu32, cu32, N, N, None, None, None)
        out32, res32 = layer32(positions, h.float(), None)
        reset_context()
    finally:
        torch.set_default_dtype(bf)
    _save("layer_c1", _layer_tensors(layer, "layer.") | dict(
        h=h, positions=positions, cu_seqlens=cu, slot_mapping=slots, out_first=out, res_first=res,
        k_rows=k_rows, v_rows=v_rows, out_first_f32=out32, res_first_f32=res32,
        d_h=h1, d_res_in=r1, d_positions=pos1, d_context_lens=ctx1, d_block_tables=bt1, d_slot_mapping=slot1,
        d_k_cache_in=kc_in, d_v_cache_in=vc_in, d_out=out_d, d_res=res_d,
        d_k_row=kc.view(-1, 256)[slot1.long()], d_v_row=vc.view(-1, 256)[slot1.long()],
        cos_sin=layer.self_attn.rotary_emb.cos_sin_cache), vars(cfg) | dict(lens=[N]))


if __name__ == "__main__":
    main()
