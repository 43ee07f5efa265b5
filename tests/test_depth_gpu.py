"""Parity IN DEPTH at GLM-4.5-Air "Air-64E" dims (r04; VERDICT r03 "Missing #3 / Weak #2").

The reference instantiates 2 layers (models/glm4_moe/model.py:21-22) and the bench times 46; the L=2 logits test cannot
tell whether the HIP path TRACKS the reference's bf16 arithmetic through a deep stack or merely stays "within bf16
noise" of it.  Two tests:

  * test_c3_model_logits_L8 — the model wrapper at L=8 (layer 0 dense, 7 MoE), decode B=4 ctx=4096: at EVERY layer
    boundary the residual stream (bf16 residual + bf16 MLP output, the sum the next add-RMSNorm forms,
    decode_layer.py:79-100 / model.py:64-78) is compared three ways — HIP vs the oracle's bf16 chain, HIP vs the
    rounding-free fp32 chain on the oracle's routing, oracle bf16 vs fp32 — so error growth is visible per layer; then
    the logits (model.py:134-145).  A row leaves the comparison at the first layer where its expert set differs from the
    oracle's (that layer must then be a near-tie of the oracle's routing).
  * test_air_dims_generate_greedy_8_steps — `nanovllm.LLM.generate()` (llm_engine.py:59-93) on an Air-dims checkpoint
    (L=3), prefill + 8 greedy decode steps in HIP-graph mode, against the oracle teacher-forced with the product's own
    tokens over a CPU paged KV cache: every generated token must be the oracle's arg-max, or lie within the measured
    bf16 logit noise of it (6 sigma, sigma = rms distance of the oracle's bf16 logits from the fp32 evaluation of the same
    step), and at most 2 of the 16 may differ at all."""
import os

import pytest
import torch

import fullsize as F
from oracle import from_product
from oracle import glm4_moe_oracle as O
from util import router_margin

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _stream(h, res):
    return h.float() + res.float()


def test_c3_model_logits_L8():
    from nanovllm import hip_ops
    from nanovllm.utils.context import reset_context, set_context
    B, ctx, L = 4, 4096, 8
    cfg = F.air_cfg(num_hidden_layers=L, num_layers_to_load=L)
    model = F.build_model(cfg, seed=13)
    model.model.embed_tokens.weight.data.normal_(0.0, 1.0, generator=torch.Generator(device="cuda").manual_seed(14))
    nkv, hd = cfg.num_key_value_heads, cfg.head_dim
    caches, bt, cl, pos, slots = F.paged_decode_inputs(B, ctx, nkv, hd, seed=31, n_layers=L)
    cpu_caches = [(k.cpu(), v.cpu()) for k, v in caches]
    for layer, (kc, vc) in zip(model.model.layers, caches):
        layer.self_attn.attn.k_cache, layer.self_attn.attn.v_cache = kc, vc
    ids = torch.randint(0, cfg.vocab_size, (B,), generator=torch.Generator().manual_seed(32))
    caps = [{} for _ in model.model.layers]
    for layer, cap in zip(model.model.layers, caps):
        if hasattr(layer.mlp, "gate"):
            layer.mlp.register_forward_pre_hook(
                lambda m, a, kw, cap=cap: cap.update(idx=kw["routing"][0].clone()), with_kwargs=True)

    # ---- HIP: the model's own forward (what the engine and the bench run), then the same layers one at a time to read
    # every boundary; the two must agree bit for bit (same kernels, same launch sequence)
    set_context(False, slot_mapping=slots.cuda(), context_lens=cl.cuda(), block_tables=bt.cuda())
    with torch.inference_mode():
        hid = model(ids.cuda(), pos.cuda())
        logits = model.compute_logits(hid)
        for layer, (kc, vc), (kc0, vc0) in zip(model.model.layers, caches, cpu_caches):  # the step wrote its K/V rows
            kc.copy_(kc0)
            vc.copy_(vc0)
        bounds = []
        hs, res = model.model.embed_tokens(ids.cuda()), None
        for layer in model.model.layers:
            hs, res = layer(pos.cuda(), hs, res, defer_combine=True)
            hm = hs.materialize() if isinstance(hs, hip_ops.DeferredCombine) else hs
            bounds.append((hm.clone(), res.clone()))
        hid2 = model.model.norm(hs)
    reset_context()
    torch.cuda.synchronize()
    assert torch.equal(hid, hid2), "layer-by-layer launch sequence differs from model.forward"
    idx_hip = [c.get("idx") for c in caps]

    # ---- oracle: bf16 chain and fp32 chain (on the bf16 chain's routing) interleaved, one layer's weights at a time ----
    cfgd, cs = from_product.cfg_dict(cfg), from_product.cos_sin_cache(cfg)
    meta = O.AttnMeta(False, slot_mapping=slots, context_lens=cl, block_tables=bt)
    embed = model.model.embed_tokens.weight.detach().cpu()
    hb, rb = torch.nn.functional.embedding(ids, embed), None
    h32, r32 = hb.float(), None
    alive = torch.ones(B, dtype=torch.bool)
    per_layer = []
    for i, layer in enumerate(model.model.layers):
        lw = from_product.layer_weights_from_module(layer)
        kc0, vc0 = cpu_caches[i]
        tr = {}
        hb, rb = O.decoder_layer(pos, hb, rb, lw, cfgd, cs, kc0.clone(), vc0.clone(), meta, trace=tr)
        h32, r32 = O.decoder_layer(pos, h32, r32, lw.to(torch.float32), cfgd, cs, kc0.float(), vc0.float(), meta,
                                   force_idx=tr.get("idx"))
        moved = []
        if "idx" in tr:
            _, _, choice = O.router_scores(tr["h_router"], lw.gate_w, lw.gate_bias)
            margin = router_margin(choice, cfg.num_experts_per_tok)
            got = idx_hip[i].cpu()
            for r in range(B):
                if alive[r] and set(got[r].tolist()) != set(tr["idx"][r].tolist()):
                    # by layer i the router input carries i layers of bf16 noise: the oracle's margin must be small
                    assert float(margin[r]) <= 0.016 * (1 + i // 2), \
                        f"layer {i} row {r}: expert set differs at routing margin {float(margin[r]):.4g}"
                    alive[r] = False
                    moved.append(r)
        rows = torch.where(alive)[0]
        assert rows.numel() >= B - 2, f"layer {i}: only {rows.numel()} of {B} rows still routed like the oracle"
        s_hip, s_ref, s_32 = _stream(*[t.cpu() for t in bounds[i]]), _stream(hb, rb), _stream(h32, r32)
        st = {"layer": i, "rows": int(rows.numel()), "moved": moved,
              "hip_vs_oracle_bf16": F.err_stats(s_hip, s_ref, rows)["rms_rel"],
              "hip_vs_fp32": F.err_stats(s_hip, s_32, rows)["rms_rel"],
              "oracle_bf16_vs_fp32": F.err_stats(s_ref, s_32, rows)["rms_rel"]}
        per_layer.append(st)
        # the HIP stream is as close to exact arithmetic as the reference's bf16 stream is, at every depth, and the two
        # bf16 evaluations are no further apart than two independent evaluations each eps from exact
        assert st["hip_vs_fp32"] <= 1.25 * st["oracle_bf16_vs_fp32"] + 1e-4, st
        assert st["hip_vs_oracle_bf16"] <= 1.5 * st["oracle_bf16_vs_fp32"] + 1e-4, st
        del lw
    normw, headw = model.model.norm.weight.detach().cpu(), model.lm_head.weight.detach().cpu()
    lref = O.linear(O.rms_forward(hb, normw, cfg.rms_norm_eps), headw)
    l32 = O.linear(O.rms_forward(h32, normw.float(), cfg.rms_norm_eps), headw.float())
    rows = torch.where(alive)[0]
    st, st32, so = F.err_stats(logits, lref, rows), F.err_stats(logits, l32, rows), F.err_stats(lref, l32, rows)
    lg, lr = logits.float().cpu()[rows], lref.float()[rows]
    top2 = torch.topk(lr, 2, dim=-1)
    margin = top2.values[:, 0] - top2.values[:, 1]
    agree = lg.argmax(-1) == top2.indices[:, 0]
    err_row = (lg - lr).abs().amax(-1)
    F.report("C3.model_logits_L8", B=B, ctx=ctx, L=L, rows_compared=int(rows.numel()), per_layer=per_layer,
             vs_oracle=st, vs_fp32=st32, oracle_vs_fp32=so, greedy_agree=int(agree.sum()),
             min_margin_where_disagree=float(margin[~agree].min()) if bool((~agree).any()) else None)
    F.assert_close_like_reference(logits, lref, l32, "logits L=8", rows, cap_factor=6.0)
    assert bool((agree | (margin <= 2 * err_row)).all()), "greedy token differs beyond the logit error"
    # error growth: the bf16 noise of the stream grows like a random walk over the layers, not geometrically
    first, last = per_layer[1]["hip_vs_fp32"], per_layer[-1]["hip_vs_fp32"]
    assert last <= 4.0 * first + 1e-3, (first, last)


# ------------------------------------------------------------------------------------------------------------------
def _write_air_checkpoint(path: str, cfg: dict, seed: int) -> dict:
    """HF key names, SURVEY 8d statistics (weights N(0, 0.02^2), norm 1 + N(0, 0.02^2), bias N(0, 0.02^2), correction
    bias U(-0.1, 0.1)); tensors drawn on the GPU, written once, returned on the CPU for the oracle."""
    import json
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    g = torch.Generator(device="cuda").manual_seed(seed)
    H, nh, nkv, hd = cfg["hidden_size"], cfg["num_attention_heads"], cfg["num_key_value_heads"], cfg["head_dim"]
    I, Id, E, V = cfg["moe_intermediate_size"], cfg["intermediate_size"], cfg["n_routed_experts"], cfg["vocab_size"]
    rn = lambda *s, std=0.02: (torch.randn(*s, generator=g, device="cuda") * std).to(BF).cpu()
    ln = lambda n: (1 + 0.02 * torch.randn(n, generator=g, device="cuda")).to(BF).cpu()
    t = {"model.embed_tokens.weight": rn(V, H, std=1.0), "model.norm.weight": ln(H), "lm_head.weight": rn(V, H)}
    for i in range(cfg["num_hidden_layers"]):
        p = f"model.layers.{i}"
        t[f"{p}.input_layernorm.weight"], t[f"{p}.post_attention_layernorm.weight"] = ln(H), ln(H)
        for name, rows in (("q_proj", nh * hd), ("k_proj", nkv * hd), ("v_proj", nkv * hd)):
            t[f"{p}.self_attn.{name}.weight"] = rn(rows, H)
            t[f"{p}.self_attn.{name}.bias"] = rn(rows)
        t[f"{p}.self_attn.o_proj.weight"] = rn(H, nh * hd)
        if i < cfg["first_k_dense_replace"]:
            t[f"{p}.mlp.gate_proj.weight"], t[f"{p}.mlp.up_proj.weight"] = rn(Id, H), rn(Id, H)
            t[f"{p}.mlp.down_proj.weight"] = rn(H, Id)
        else:
            t[f"{p}.mlp.gate.weight"] = rn(E, H)
            t[f"{p}.mlp.gate.e_score_correction_bias"] = (torch.rand(E, generator=g, device="cuda") * 0.2 - 0.1).float().cpu()
            for e in list(range(E)) + ["shared"]:
                q = f"{p}.mlp.shared_experts" if e == "shared" else f"{p}.mlp.experts.{e}"
                t[f"{q}.gate_proj.weight"], t[f"{q}.up_proj.weight"] = rn(I, H), rn(I, H)
                t[f"{q}.down_proj.weight"] = rn(H, I)
    save_file({k: v.contiguous() for k, v in t.items()}, os.path.join(path, "model.safetensors"))
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cfg, f)
    return t


def test_air_dims_generate_greedy_8_steps(tmp_path):
    from nanovllm import LLM, SamplingParams
    from util import oracle_layers_from_checkpoint
    L, steps = 3, 8
    cfg = dict(F.AIR64E, model_type="glm4_moe", num_hidden_layers=L, num_layers_to_load=L, torch_dtype="bfloat16")
    path = str(tmp_path / "air_l3")
    t = _write_air_checkpoint(path, cfg, seed=21)
    g = torch.Generator().manual_seed(22)
    prompts = [torch.randint(0, cfg["vocab_size"], (n,), generator=g).tolist() for n in (19, 7)]
    llm = LLM(path, enforce_eager=False, max_model_len=1024, max_num_batched_tokens=2048, max_num_seqs=4,
              num_kvcache_blocks=16)
    try:
        out = llm.generate(prompts, SamplingParams(temperature=0.0, max_tokens=steps, ignore_eos=True), use_tqdm=False)
    finally:
        llm.exit()
    torch.cuda.empty_cache()
    got = [o["token_ids"] for o in out]
    # ---- oracle, teacher-forced with the product's tokens, over a CPU paged cache (one 256-token block per sequence) ----
    layers = oracle_layers_from_checkpoint(t, cfg)
    cs = O.build_cos_sin_cache(O.rotary_dim_of(cfg["head_dim"], cfg["partial_rotary_factor"]),
                               cfg["max_position_embeddings"], cfg["rope_theta"])
    nkv, hd = cfg["num_key_value_heads"], cfg["head_dim"]
    embed, normw, headw = t["model.embed_tokens.weight"], t["model.norm.weight"], t["lm_head.weight"]
    agree = total = 0
    worst, worst_sigma = 0.0, 0.0
    layers32 = [l.to(torch.float32) for l in layers]
    e32, n32, h32w = embed.float(), normw.float(), headw.float()

    def both(ids, pos, kvs, kvs32, meta, last):
        """The oracle's bf16 logits and the rounding-free fp32 logits of the same step (own fp32 K/V cache)."""
        h = O.model_forward(ids, pos, embed, layers, normw, cfg, cs, kvs, meta)
        hf = O.model_forward(ids, pos, e32, layers32, n32, cfg, cs, kvs32, meta)
        return O.linear(h[last:], headw)[0].float(), O.linear(hf[last:], h32w)[0]

    for prompt, toks in zip(prompts, got):
        assert len(toks) == steps
        kvs = [(torch.zeros(1, 256, nkv, hd, dtype=BF), torch.zeros(1, 256, nkv, hd, dtype=BF)) for _ in range(L)]
        kvs32 = [(torch.zeros(1, 256, nkv, hd), torch.zeros(1, 256, nkv, hd)) for _ in range(L)]
        n = len(prompt)
        cu = torch.tensor([0, n], dtype=torch.int32)
        meta = O.AttnMeta(True, slot_mapping=torch.arange(n, dtype=torch.int32), cu_seqlens_q=cu, cu_seqlens_k=cu)
        logits, l32 = both(torch.tensor(prompt), torch.arange(n), kvs, kvs32, meta, -1)
        for s, tok in enumerate(toks):
            top = int(logits.argmax())
            total += 1
            if top == tok:
                agree += 1
            else:
                # The product picked `tok`, the oracle `top`.  Each side's logits are the exact ones plus that side's bf16
                # noise; sigma = rms(oracle bf16 - fp32 logits) of THIS step measures it (the HIP path's equals the
                # oracle's: test_c3_model_logits_L8).  The oracle's gap between the two tokens is then the difference of
                # four such errors (std 2 sigma) at most: 6 sigma is three standard deviations of it.
                gap = float(logits[top] - logits[tok])
                sigma = float((logits - l32).pow(2).mean().sqrt())
                worst, worst_sigma = max(worst, gap), max(worst_sigma, gap / sigma)
                assert gap <= 6 * sigma, f"step {s}: token {tok} vs oracle {top}, oracle logit gap {gap:.3f} = {gap / sigma:.1f} sigma"
            if s + 1 == len(toks):
                break
            p = n + s  # position of the token just generated
            meta = O.AttnMeta(False, slot_mapping=torch.tensor([p], dtype=torch.int32),
                              context_lens=torch.tensor([p + 1], dtype=torch.int32),
                              block_tables=torch.tensor([[0]], dtype=torch.int32))
            logits, l32 = both(torch.tensor([tok]), torch.tensor([p]), kvs, kvs32, meta, 0)
    F.report("C3dims.generate_greedy_L3", prompts=[len(p) for p in prompts], steps=steps, tokens_compared=total,
             tokens_equal_to_oracle_argmax=agree, worst_gap_where_different=worst, worst_gap_in_sigma=worst_sigma)
    assert agree >= total - 2, f"{agree}/{total} greedy tokens equal the oracle's"
