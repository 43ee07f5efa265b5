"""FULL-SIZE parity: the instantiations bench.py times, compared with the CPU oracle at the BASELINE configs' own
dimensions (GLM-4.5-Air "Air-64E": H=4096, 96/8 heads x 128, E=64 top-8 + shared, I=1408, dense I=10944).

    C3  bs=32, ctx=4096  (the config the metric is quoted on)   -> test_c3_*
    C2  bs=1,  ctx=1024  (greedy single-sequence decode)         -> test_c2_*

Every test appends its observed error figures to gpurun_out/parity_fullsize.jsonl (copied to profiles/ per round):
HIP vs the reference's bf16 arithmetic (oracle), HIP vs the rounding-free fp32 evaluation of the same algorithm, and
the oracle's own distance from that fp32 evaluation.  Acceptance = the reference scripts' own rtol = atol = 1e-2
(nanovllm/tests/moe.py:40, tests/mlp.py:65) on every row whose expert set agrees; rows whose set differs must be
near-ties of the reference routing (margin below 2 bf16 ulp of a score), and on the product's own router input the
expert sets must be bit-exact up to PROVEN logit-rounding flips / fp32 ties (fullsize.router_proof).
Reference path: models/glm4_moe/decode_layer.py:79-100."""
import pytest
import torch

import fullsize as F
from oracle import glm4_moe_oracle as O
from util import assert_close_bf16, router_margin

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    from nanovllm import hip_ops
    hip_ops.load_library()
    return hip_ops


def _layer_decode_case(ops, tag: str, B: int, ctx: int, layer_idx: int, pad_rows: int = 0, **cfg_over):
    from nanovllm.utils.context import reset_context, set_context
    cfg = F.air_cfg(**cfg_over)
    layer = F.build_layer(cfg, layer_idx, seed=0)
    nkv, hd, H = cfg.num_key_value_heads, cfg.head_dim, cfg.hidden_size
    caches, bt, cl, pos, slots = F.paged_decode_inputs(B, ctx, nkv, hd, seed=7, pad_rows=pad_rows)
    kc, vc = caches[0]
    kc0, vc0 = kc.cpu(), vc.cpu()
    Bp = B + pad_rows
    h, res = F.randn_bf16(Bp, H, seed=1), F.randn_bf16(Bp, H, seed=2)
    layer.self_attn.attn.k_cache, layer.self_attn.attn.v_cache = kc, vc
    cap = {}
    is_moe = hasattr(layer.mlp, "gate")
    if is_moe:
        layer.mlp.register_forward_pre_hook(
            lambda m, a, kw: cap.update(h_router=a[0], routing=kw.get("routing")), with_kwargs=True)
    set_context(False, slot_mapping=slots.cuda(), context_lens=cl.cuda(), block_tables=bt.cuda())
    with torch.inference_mode():
        out, res_out = layer(pos.cuda(), h, res)
    reset_context()
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all() and torch.isfinite(res_out.float()).all()

    # ---- oracle: the reference's bf16 arithmetic, then the rounding-free fp32 evaluation on the same routing ----
    lw, cfgd, cs = F.oracle_layer_inputs(layer, cfg)
    meta = O.AttnMeta(False, slot_mapping=slots[:B], context_lens=cl[:B], block_tables=bt[:B])
    hc, rc = h.cpu()[:B], res.cpu()[:B]
    kc_ref, vc_ref = kc0.clone(), vc0.clone()
    tr = {}
    ref_out, ref_res = O.decoder_layer(pos[:B], hc, rc, lw, cfgd, cs, kc_ref, vc_ref, meta, trace=tr)
    tr32 = {}
    out32, res32 = O.decoder_layer(pos[:B], hc.float(), rc.float(), lw.to(torch.float32), cfgd, cs, kc0.float(),
                                   vc0.float(), meta, force_idx=tr.get("idx"), trace=tr32)

    # ---- KV cache: the new rows, and nothing else touched ----
    D = nkv * hd
    s = slots[:B].long()
    kg, vg = kc.cpu().view(-1, D), vc.cpu().view(-1, D)
    st_k, st_v = F.err_stats(kg[s], kc_ref.view(-1, D)[s]), F.err_stats(vg[s], vc_ref.view(-1, D)[s])
    untouched = torch.ones(kg.shape[0], dtype=torch.bool)
    untouched[s] = False
    assert torch.equal(kg[untouched], kc0.view(-1, D)[untouched]) and torch.equal(vg[untouched], vc0.view(-1, D)[untouched])
    st_res = F.err_stats(res_out[:B], ref_res)

    # ---- routing ----
    routing = {}
    rows = None
    if is_moe:
        idx_f, w_f = cap["routing"]
        gate = layer.mlp.gate
        routing = F.router_proof(ops, cap["h_router"], gate.weight.data, gate.e_score_correction_bias, gate.top_k,
                                 float(gate.routed_scaling_factor), idx_f, w_f, n_valid=B)
        _, _, choice = O.router_scores(tr["h_router"], lw.gate_w, lw.gate_bias)
        rows, moved = F.same_set_rows(idx_f[:B].cpu(), tr["idx"], router_margin(choice, gate.top_k), 0.016)
        routing["rows_moved_by_upstream_rounding"] = len(moved)
        # observed (profiles/r02/parity_fullsize.jsonl): 0/32, 0/1, 0/64, 1/256 — each a near-tie (margin <= 0.016, checked above)
        assert len(moved) <= max(1, -(-2 * B // 100)), f"{tag}: {len(moved)}/{B} rows routed differently"

    st_out = F.err_stats(out[:B], ref_out, rows)
    st_out32 = F.err_stats(out[:B], out32, rows)
    st_ref32 = F.err_stats(ref_out, out32, rows)
    F.report(tag, B=B, ctx=ctx, layer_idx=layer_idx, pad_rows=pad_rows, k_rows=st_k, v_rows=st_v, residual=st_res,
             out_vs_oracle_bf16=st_out, out_vs_fp32=st_out32, oracle_bf16_vs_fp32=st_ref32, routing=routing)

    if pad_rows:  # graph padding rows (slot -1, context_len 0) must come out finite; they are never read
        assert torch.isfinite(out[B:].float()).all()
    # V rows: qkv GEMM + bias only (one fp32 reduction); K rows: the same through the rotary
    assert_close_bf16(vg[s], vc_ref.view(-1, D)[s], 1, 0.05, atol=2 ** -9, what=f"{tag} v rows")
    assert_close_bf16(kg[s], kc_ref.view(-1, D)[s], 2, 0.05, atol=2 ** -5, what=f"{tag} k rows")
    # residual = bf16(o_proj + residual_in): the small o_proj term (attention P rounded to bf16 in the kernel, fp32 in the
    # stand-in oracle) flips the rounding of the sum by one ulp on a few per cent of the elements
    assert_close_bf16(res_out[:B], ref_res, 1, 0.10, atol=2 ** -7, what=f"{tag} residual")
    F.assert_close_like_reference(out[:B], ref_out, out32, f"{tag} out", rows)


def test_c3_moe_layer_decode(ops):
    """(i) one MoE decoder layer, decode B=32 ctx=4096 over a permuted block table."""
    _layer_decode_case(ops, "C3.layer_moe_decode", 32, 4096, 1)


def test_c2_moe_layer_decode(ops):
    """(ii) bs=1 ctx=1024: the R<=16 gate|up configuration and the split + reduce attention."""
    _layer_decode_case(ops, "C2.layer_moe_decode", 1, 1024, 1)


def test_c2_dense_layer_decode(ops):
    """(ii) layer 0 (dense MLP, I=10944) at bs=1 ctx=1024: the dense down-projection configuration."""
    _layer_decode_case(ops, "C2.layer_dense_decode", 1, 1024, 0)


def test_c5_tp4_rank_layer_fp8_experts(ops):
    """BASELINE config 5 (fp8 weight-quantised experts, TP=4, bs=64, seq=8192) as ONE RANK sees it: 24 q / 2 kv heads,
    expert intermediate 1408/4 = 352, fp8 e4m3 experts, 64 sequences at context 8192 (32 blocks each, permuted).  The
    collectives are not part of this (TP sharding is covered by tests/test_tp_gloo_cpu.py); the oracle evaluates the
    reference's arithmetic on the same dequantised bytes."""
    _layer_decode_case(ops, "C5.tp4_rank_layer_fp8_decode", 64, 8192, 1, num_attention_heads=24, num_key_value_heads=2,
                       moe_intermediate_size=352, intermediate_size=2736, expert_weight_dtype="fp8_e4m3",
                       max_position_embeddings=8192)


def test_c4_tp8_rank_layer_bs256(ops):
    """BASELINE config 4 (TP=8, bs=256) as one rank's tensor-parallel layer: 12 q heads / 1 kv head, expert
    intermediate 176, 256 sequences at context 4096."""
    _layer_decode_case(ops, "C4.tp8_rank_layer_decode_bs256", 256, 4096, 1, num_attention_heads=12,
                       num_key_value_heads=1, moe_intermediate_size=176, intermediate_size=1368)


def test_c4_expert_parallel_moe_bs256(ops):
    """BASELINE config 4's MoE block with expert-parallel all-to-all at Air dims: 8 logical ranks x 8 experts (full
    intermediate size), 256 tokens, dispatch -> experts -> combine (distributed/expert_parallel.py, collectives done by
    copies on one device) against the oracle's single-rank MoE and against the TP=1 kernels."""
    from nanovllm.distributed import expert_parallel as EP
    cfg = F.air_cfg()
    H, E, k, I = cfg.hidden_size, cfg.n_routed_experts, cfg.num_experts_per_tok, cfg.moe_intermediate_size
    T, ep = 256, 8
    x, idx, w, offs, tok, scale, pos = _bench_routing(ops, cfg, T, seed=501)
    g = torch.Generator(device="cuda").manual_seed(9)
    w13 = torch.empty(E + 1, 2 * I, H, device="cuda", dtype=BF).normal_(0.0, 0.02, generator=g)
    w2 = torch.empty(E + 1, H, I, device="cuda", dtype=BF).normal_(0.0, 0.02, generator=g)
    R = T * k + T
    act = ops.grouped_gemm_gateup_silu(x, w13, offs, tok, R)
    y = ops.grouped_gemm_down(act, w2, offs, scale, E)
    ref_dev = ops.moe_combine(y, pos, T, k, T * k)
    El = E // ep

    def shared(hh):
        a = ops.grouped_gemm_gateup_silu(hh, w13[E:E + 1], None, None, hh.shape[0])
        return ops.grouped_gemm_down(a, w2[E:E + 1], None, None, 0)
    gens = [EP.EPExperts(ops, ep, r, E, k, w13[r * El:(r + 1) * El], w2[r * El:(r + 1) * El], shared).forward(x, idx, w)
            for r in range(ep)]
    outs = EP.run_emulated(gens)
    for o in outs:
        assert torch.equal(o, outs[0])
    ref = O.moe(x.cpu(), None, None, k, 1.0, w13[:E].cpu(), w2[:E].cpu(), w13[E].cpu(), w2[E].cpu(),
                topk=(idx.cpu(), w.cpu()))
    st_ep, st_tp1 = F.err_stats(outs[0], ref), F.err_stats(ref_dev, ref)
    same = bool(torch.equal(outs[0], ref_dev))
    F.report("C4.expert_parallel_moe_T256_ep8", ep_vs_oracle=st_ep, tp1_vs_oracle=st_tp1, ep_equals_tp1_bitwise=same,
             ep_vs_tp1=F.err_stats(outs[0], ref_dev))
    F.assert_allclose_ref_tol(outs[0], ref, "EP=8 MoE block vs oracle", outlier_frac=1e-3)
    if not same:  # another K-split configuration on ranks with few rows: single ulps (atol for sums that cancel)
        assert_close_bf16(outs[0], ref_dev, 2, 0.05, atol=2 ** -5, what="EP=8 vs TP=1 kernels")


def _layer_prefill_case(ops, tag: str, seqlens, layer_idx: int, **cfg_over):
    """One decoder layer over a ragged prefill batch at Air dims (varlen causal attention, new K/V rows stored through a
    permuted slot mapping, large-M GEMMs, prefill router + bucketing), against the oracle's prefill branch
    (layers/attention.py:85-100, decode_layer.py:79-100)."""
    from nanovllm.utils.context import reset_context, set_context
    cfg = F.air_cfg(**cfg_over)
    layer = F.build_layer(cfg, layer_idx, seed=0)
    nkv, hd, H = cfg.num_key_value_heads, cfg.head_dim, cfg.hidden_size
    T = int(sum(seqlens))
    cu = torch.tensor([0] + list(seqlens), dtype=torch.int32).cumsum(0).to(torch.int32)
    pos = torch.cat([torch.arange(n) for n in seqlens]).to(torch.int64)
    bs = 256
    nblk = sum((n + bs - 1) // bs for n in seqlens) + 1
    g = torch.Generator().manual_seed(11)
    perm = torch.randperm(nblk, generator=g)
    slots, b0 = [], 0
    for n in seqlens:  # each sequence owns its own (permuted) blocks
        nb = (n + bs - 1) // bs
        blocks = perm[b0:b0 + nb]
        b0 += nb
        t = torch.arange(n)
        slots.append(blocks[t // bs] * bs + t % bs)
    slots = torch.cat(slots).to(torch.int32)
    kc = torch.zeros(nblk, bs, nkv, hd, dtype=BF, device="cuda")
    vc = torch.zeros_like(kc)
    layer.self_attn.attn.k_cache, layer.self_attn.attn.v_cache = kc, vc
    h, res = F.randn_bf16(T, H, seed=1), F.randn_bf16(T, H, seed=2)
    cap = {}
    is_moe = hasattr(layer.mlp, "gate")
    if is_moe:
        layer.mlp.register_forward_pre_hook(
            lambda m, a, kw: cap.update(h_router=a[0], routing=kw.get("routing")), with_kwargs=True)
        layer.mlp.gate.register_forward_hook(lambda m, i, o: cap.update(gate_out=o, gate_in=i[0]))
    set_context(True, cu.cuda(), cu.cuda(), max(seqlens), max(seqlens), slots.cuda(), None, None)
    with torch.inference_mode():
        out, res_out = layer(pos.cuda(), h, res)
    reset_context()
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all() and torch.isfinite(res_out.float()).all()

    lw, cfgd, cs = F.oracle_layer_inputs(layer, cfg)
    meta = O.AttnMeta(True, slot_mapping=slots, cu_seqlens_q=cu, cu_seqlens_k=cu)
    hc, rc = h.cpu(), res.cpu()
    kz = torch.zeros(nblk, bs, nkv, hd, dtype=BF)
    kc_ref, vc_ref = kz.clone(), kz.clone()
    tr = {}
    ref_out, ref_res = O.decoder_layer(pos, hc, rc, lw, cfgd, cs, kc_ref, vc_ref, meta, trace=tr)
    out32, res32 = O.decoder_layer(pos, hc.float(), rc.float(), lw.to(torch.float32), cfgd, cs, kz.float(), kz.float(),
                                   meta, force_idx=tr.get("idx"))
    D = nkv * hd
    sl = slots.long()
    kg, vg = kc.cpu().view(-1, D), vc.cpu().view(-1, D)
    untouched = torch.ones(kg.shape[0], dtype=torch.bool)
    untouched[sl] = False
    assert not kg[untouched].any() and not vg[untouched].any(), f"{tag}: cache rows outside the slot mapping written"
    st_k, st_v = F.err_stats(kg[sl], kc_ref.view(-1, D)[sl]), F.err_stats(vg[sl], vc_ref.view(-1, D)[sl])
    st_res = F.err_stats(res_out, ref_res)
    routing, rows = {}, None
    if is_moe:
        gate = layer.mlp.gate
        if cap.get("routing") is not None:
            idx_f, w_f = cap["routing"]
            h_router = cap["h_router"]
        else:  # un-fused path: the gate module ran on its own input
            idx_f, w_f = cap["gate_out"][0], cap["gate_out"][1]
            h_router = cap["gate_in"]
        routing = F.router_proof(ops, h_router, gate.weight.data, gate.e_score_correction_bias, gate.top_k,
                                 float(gate.routed_scaling_factor), idx_f, w_f, n_valid=T)
        _, _, choice = O.router_scores(tr["h_router"], lw.gate_w, lw.gate_bias)
        rows, moved = F.same_set_rows(idx_f.cpu(), tr["idx"], router_margin(choice, gate.top_k), 0.016)
        routing["rows_moved_by_upstream_rounding"] = len(moved)
        # observed: 9/1024 near-ties moved by upstream rounding
        assert len(moved) <= max(1, -(-2 * T // 100)), f"{tag}: {len(moved)}/{T} rows routed differently"
    st_out = F.err_stats(out, ref_out, rows)
    st_out32 = F.err_stats(out, out32, rows)
    st_ref32 = F.err_stats(ref_out, out32, rows)
    F.report(tag, T=T, seqlens=list(seqlens), layer_idx=layer_idx, k_rows=st_k, v_rows=st_v, residual=st_res,
             residual_vs_fp32=F.err_stats(res_out, res32), oracle_residual_bf16_vs_fp32=F.err_stats(ref_res, res32),
             out_vs_oracle_bf16=st_out, out_vs_fp32=st_out32, oracle_bf16_vs_fp32=st_ref32, routing=routing)
    assert_close_bf16(vg[sl], vc_ref.view(-1, D)[sl], 1, 0.05, atol=2 ** -9, what=f"{tag} v rows")
    assert_close_bf16(kg[sl], kc_ref.view(-1, D)[sl], 2, 0.05, atol=2 ** -5, what=f"{tag} k rows")
    # residual = bf16(a + r), a = the o_proj output, r = the residual input (exact).  With few keys per query the attention
    # output is not averaged down as at ctx = 4096: |a| ~ 1, and a — a K = 12288 sum with cancellation — differs between two
    # bf16 evaluations by ~2^-9 of its ROW rms (attention P in bf16 here, fp32 in the stand-in oracle), not by ulps of the
    # element.  Same acceptance as for the layer output: as close to the reference's bf16 result as that is to exact.
    F.assert_close_like_reference(res_out, ref_res, res32, f"{tag} residual")
    F.assert_close_like_reference(out, ref_out, out32, f"{tag} out", rows)


def test_c3_moe_layer_prefill_ragged(ops):
    """Prefill side of config 3 at Air dims: three ragged prompts (1, 383 and 640 tokens = 1024 rows, 8192 + 1024 expert
    rows through the large-M grouped GEMMs), block-permuted slot mapping."""
    _layer_prefill_case(ops, "C3.layer_moe_prefill_ragged", (383, 1, 640), 1)


def test_c3_dense_layer_prefill(ops):
    """Layer 0 (dense MLP, I=10944) over one 300-token prompt."""
    _layer_prefill_case(ops, "C3.layer_dense_prefill", (300,), 0)


def test_graph_bucket_padding_rows(ops):
    """bs=3 in a 4-bucket at Air dims: the padding row (slot -1, context_len 0, block table -1) goes through the fused
    decode attention (paged_attn.hip) without touching the cache or producing NaNs; the real rows still match."""
    _layer_decode_case(ops, "pad.layer_moe_decode_3of4", 3, 777, 1, pad_rows=1)


# ------------------------------------------------------------------------------------------------------------------
def _bench_routing(ops, cfg, T, seed):
    """x, router weights and the sorted-row structure exactly as bench.py's kernel measurements build them."""
    H, E, k = cfg.hidden_size, cfg.n_routed_experts, cfg.num_experts_per_tok
    x = F.randn_bf16(T, H, seed=seed)
    gw = F.randn_bf16(E, H, seed=seed + 1, scale=0.02)
    bias = (torch.rand(E, generator=torch.Generator().manual_seed(seed + 2)) * 0.2 - 0.1).float().cuda()
    idx, w = ops.router_topk(x, gw, bias, k, 1.0)
    offs, tok, scale, pos = ops.moe_align(idx, w, E, True)
    return x, idx, w, offs, tok, scale, pos


def _check_groups(tag, x, w13, w2, offs, tok, scale, act, y, E, groups, row_sel):
    """Per-expert rows of the two grouped GEMMs against the oracle's expert MLP pieces (mlp.py:49-54, moe.py:149)."""
    offs_c, tok_c, scale_c = offs.cpu(), tok.cpu(), scale.cpu()
    worst = {"act": None, "y": None}
    for e in groups:
        lo, hi = int(offs_c[e]), int(offs_c[e + 1])
        if lo == hi:
            continue
        rr = row_sel(lo, hi)                           # slice or CPU index tensor
        rg = rr.cuda() if torch.is_tensor(rr) else rr
        xe = x[tok[rg].long()].cpu()
        a_ref = O.silu_and_mul(O.linear(xe, w13[e].cpu()))
        a_got = act[rg].cpu()
        # gate, up and bf16(silu) can each flip one ulp through the fp32 summation order: <= 3 ulp stacked.  Where the
        # gate is well below zero silu is ill-conditioned in ulp terms (one ulp of g = -4 moves silu(g) by 3.4 of ITS
        # ulps, g = -6 by 6): such elements are tiny in absolute terms, |d act| <= ulp(g) |silu'(g)| |u| < 2^-7
        assert_close_bf16(a_got, a_ref, 3, 0.06, atol=2 ** -7, what=f"{tag} act group {e}")
        y_ref = O.linear(a_got, w2[e].cpu())          # the product's own act rows: isolates the down GEMM
        if e < E:
            y_ref = y_ref * scale_c[rr].unsqueeze(-1)  # bf16 multiply by the routing weight (moe.py:149)
        assert_close_bf16(y[rg].cpu(), y_ref, 1, 0.05, atol=2 ** -9, what=f"{tag} y group {e}")
        for name, got, ref in (("act", a_got, a_ref), ("y", y[rg].cpu(), y_ref)):
            st = F.err_stats(got, ref)
            if worst[name] is None or st["max_ulp"] > worst[name]["max_ulp"] or st["frac_inexact"] > worst[name]["frac_inexact"]:
                worst[name] = st
    return worst


def test_c3_grouped_gemms_decode_shape(ops):
    """(iii) grouped gate|up+SiLU and down at H=4096, I=1408, G=65 with the bench's routing at T=32 — the
    gemm_smallm LDS-DMA instantiations — every group, every row."""
    cfg = F.air_cfg()
    H, E, k, I = cfg.hidden_size, cfg.n_routed_experts, cfg.num_experts_per_tok, cfg.moe_intermediate_size
    T = 32
    x, idx, w, offs, tok, scale, pos = _bench_routing(ops, cfg, T, seed=101)
    w13 = ops.padded_weight(2 * I, H, E + 1, dtype=BF, device="cuda")
    w2 = ops.padded_weight(H, I, E + 1, dtype=BF, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(5)
    w13.normal_(0.0, 0.02, generator=g)
    w2.normal_(0.0, 0.02, generator=g)
    R = T * k + T
    act = ops.grouped_gemm_gateup_silu(x, w13, offs, tok, R)
    y = ops.grouped_gemm_down(act, w2, offs, scale, E)
    torch.cuda.synchronize()
    worst = _check_groups("C3 decode", x, w13, w2, offs, tok, scale, act, y, E, range(E + 1),
                          lambda lo, hi: slice(lo, hi))
    touched = int((offs.cpu()[1:E + 1] - offs.cpu()[:E] > 0).sum())
    # single sequence (C2): R <= 16 configuration
    x1, idx1, w1, offs1, tok1, scale1, pos1 = _bench_routing(ops, cfg, 1, seed=111)
    act1 = ops.grouped_gemm_gateup_silu(x1, w13, offs1, tok1, k + 1)
    y1 = ops.grouped_gemm_down(act1, w2, offs1, scale1, E)
    worst1 = _check_groups("C2 decode", x1, w13, w2, offs1, tok1, scale1, act1, y1, E, range(E + 1),
                           lambda lo, hi: slice(lo, hi))
    F.report("C3.grouped_gemms_T32", touched_groups=touched, worst=worst)
    F.report("C2.grouped_gemms_T1", worst=worst1)


def test_c3_grouped_gemms_prefill_shape(ops):
    """(iii) the MFMA-bound instantiation bench.py times (gemm_largem, T=16384 -> 147456 rows): sampled row ranges of
    sampled groups (first rows, and the partial last row block) against the oracle, then — over ALL rows — the
    size-independent property that the MoE block is permutation-equivariant bit for bit (a token's result may not
    depend on where its rows land inside the expert groups / tiles)."""
    cfg = F.air_cfg()
    H, E, k, I = cfg.hidden_size, cfg.n_routed_experts, cfg.num_experts_per_tok, cfg.moe_intermediate_size
    T = 16384
    x, idx, w, offs, tok, scale, pos = _bench_routing(ops, cfg, T, seed=201)
    w13 = ops.padded_weight(2 * I, H, E + 1, dtype=BF, device="cuda")
    w2 = ops.padded_weight(H, I, E + 1, dtype=BF, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(6)
    w13.normal_(0.0, 0.02, generator=g)
    w2.normal_(0.0, 0.02, generator=g)
    R = T * k + T
    act = ops.grouped_gemm_gateup_silu(x, w13, offs, tok, R)
    y = ops.grouped_gemm_down(act, w2, offs, scale, E)
    out = ops.moe_combine(y, pos, T, k, T * k)
    torch.cuda.synchronize()

    def sel(lo, hi):  # first 384 rows + the last 200 (covers the group's partial last 256-row block)
        n = hi - lo
        return torch.cat([torch.arange(0, min(384, n)), torch.arange(max(0, n - 200), n)]).unique() + lo
    worst = _check_groups("C3 prefill16k", x, w13, w2, offs, tok, scale, act, y, E, (0, 31, 63, 64), sel)
    F.report("C3.grouped_gemms_T16384", worst=worst, rows=R)
    # permutation equivariance over all 16384 tokens, bit for bit
    perm = torch.randperm(T, generator=torch.Generator().manual_seed(9)).cuda()
    xp = x[perm].contiguous()
    idx2, w2r = idx[perm].contiguous(), w[perm].contiguous()
    offs2, tok2, scale2, pos2 = ops.moe_align(idx2, w2r, E, True)
    assert torch.equal(offs2, offs)
    act2 = ops.grouped_gemm_gateup_silu(xp, w13, offs2, tok2, R)
    y2 = ops.grouped_gemm_down(act2, w2, offs2, scale2, E)
    out2 = ops.moe_combine(y2, pos2, T, k, T * k)
    assert torch.equal(out2, out[perm]), "MoE block is not permutation-equivariant at T=16384"


@pytest.mark.parametrize("T", [64, 4096])
def test_c5_fp8_expert_gemms_full_size(ops, T):
    """N4 / BASELINE config 5 flavour at Air dims (H=4096, I=1408, G=65): fp8 e4m3 weight-only experts.  T=64 (the
    config's batch) runs the streaming fp8 kernel, T=4096 the prefill path (exact fp8->bf16 expansion + tiled MFMA
    kernel with the row scale on the accumulator).  Oracle = the reference's expert arithmetic on the dequantised
    weights with the SAME bytes and scales (oracle.linear_fp8w): sampled rows of sampled groups."""
    cfg = F.air_cfg()
    H, E, k, I = cfg.hidden_size, cfg.n_routed_experts, cfg.num_experts_per_tok, cfg.moe_intermediate_size
    x, idx, w, offs, tok, scale, pos = _bench_routing(ops, cfg, T, seed=401)
    g = torch.Generator(device="cuda").manual_seed(8)
    w13 = torch.empty(E + 1, 2 * I, H, device="cuda", dtype=BF).normal_(0.0, 0.02, generator=g)
    w2 = torch.empty(E + 1, H, I, device="cuda", dtype=BF).normal_(0.0, 0.02, generator=g)
    q13, s13 = ops.quantize_fp8_rowwise(w13)
    q2, s2 = ops.quantize_fp8_rowwise(w2)
    del w13, w2
    R = T * k + T
    act = ops.grouped_gemm_gateup_silu_fp8w(x, q13, s13, offs, tok, R)
    y = ops.grouped_gemm_down_fp8w(act, q2, s2, offs, scale, E)
    torch.cuda.synchronize()
    offs_c, tok_c, scale_c = offs.cpu(), tok.cpu(), scale.cpu()
    worst = {}
    for e in (0, 31, 63, 64):
        lo, hi = int(offs_c[e]), int(offs_c[e + 1])
        n = hi - lo
        if n == 0:
            continue
        rr = (torch.cat([torch.arange(0, min(256, n)), torch.arange(max(0, n - 150), n)]).unique() + lo)
        rg = rr.cuda()
        qe13 = q13[e].cpu().contiguous().view(torch.float8_e4m3fn)
        qe2 = q2[e].cpu().contiguous().view(torch.float8_e4m3fn)
        a_ref = O.silu_and_mul(O.linear_fp8w(x[tok[rg].long()].cpu(), qe13, s13[e].cpu()))
        a_got = act[rg].cpu()
        assert_close_bf16(a_got, a_ref, 3, 0.06, atol=2 ** -7, what=f"fp8 T={T} act group {e}")
        y_ref = O.linear_fp8w(a_got, qe2, s2[e].cpu())
        if e < E:
            y_ref = y_ref * scale_c[rr].unsqueeze(-1)
        assert_close_bf16(y[rg].cpu(), y_ref, 1, 0.05, atol=2 ** -9, what=f"fp8 T={T} y group {e}")
        worst[f"group{e}"] = {"act": F.err_stats(a_got, a_ref), "y": F.err_stats(y[rg].cpu(), y_ref)}
    F.report(f"C5.fp8_expert_gemms_T{T}", rows=R, worst=worst)


def test_quantize_fp8_rows(ops):
    """glm_quantize_fp8_rows (dynamic per-row e4m3 quantisation of the activations of the fp8 x fp8 expert GEMM) against its
    restatement in the oracle: scales bit-equal, bytes bit-equal (fp32 division, RNE, clamp at +-448)."""
    for T, H, seed in ((5, 4096, 1), (300, 1408, 2), (64, 128, 3)):
        x = F.randn_bf16(T, H, seed=seed, scale=3.0)
        x[0, : H // 2] = 0
        if T > 2:
            x[2] = 0
        q, s = ops.quantize_fp8_rows(x)
        qr, sr = O.quantize_fp8_rows(x.cpu())
        assert torch.equal(s.cpu(), sr), "row scales differ"
        assert torch.equal(q.cpu(), qr.view(torch.uint8)), f"{int((q.cpu() != qr.view(torch.uint8)).sum())} fp8 bytes differ"


@pytest.mark.parametrize("T", [4096, 16384])
def test_c5_fp8a8_expert_gemms_full_size(ops, T):
    """r05, BASELINE configs[4] ("fp8 weight-quantised experts on CDNA4 fp8 MFMA") at Air dims (H=4096, I=1408, G=65), OPT-IN
    W8A8: e4m3 x e4m3 on v_mfma_scale_f32_16x16x128_f8f6f4.  gate|up with a per-token activation scale (block scales 2^0)
    and the MX quantisation of act in its epilogue; down with the act rows' E8M0 block scales inside the MFMA.
    PARITY UNPINNED (the reference has no fp8 path: layers/linear_awq_new.py:254-255 is int4 dequant-then-GEMM); the oracle
    leg is the reference's expert arithmetic (mlp.py:49-54, moe.py:149) on the SAME dequantised bytes, activations
    included (oracle.gateup_silu_fp8a8 / down_fp8a8, the quantisers restated there).  Sampled rows of sampled groups:
      * act (dequantised) vs the oracle's act BEFORE its MX quantisation: within one e4m3 step of the block (2^-3 relative
        to the block maximum) + the bf16 rounding-order slack of the weight-only test;
      * the act BYTES and scale bytes vs the oracle's quantiser applied to the oracle's act: equal wherever the two bf16
        act values are equal (counted), since the quantiser itself is exact arithmetic;
      * y vs the oracle's down projection fed with the KERNEL's act bytes: <= 1 bf16 ulp on a few elements.
    Reported next to it: the distance of the W8A8 result to the weight-only (bf16-activation) oracle on the same rows."""
    cfg = F.air_cfg()
    H, E, k, I = cfg.hidden_size, cfg.n_routed_experts, cfg.num_experts_per_tok, cfg.moe_intermediate_size
    x, idx, w, offs, tok, scale, pos = _bench_routing(ops, cfg, T, seed=431)
    g = torch.Generator(device="cuda").manual_seed(9)
    w13 = torch.empty(E + 1, 2 * I, H, device="cuda", dtype=BF).normal_(0.0, 0.02, generator=g)
    w2 = torch.empty(E + 1, H, I, device="cuda", dtype=BF).normal_(0.0, 0.02, generator=g)
    q13, s13 = ops.quantize_fp8_rowwise(w13)
    q2, s2 = ops.quantize_fp8_rowwise(w2)
    del w13, w2
    R = T * k + T
    xq, xs = ops.quantize_fp8_rows(x)
    r = ops.grouped_gemm_gateup_silu_fp8a8(xq, xs, q13, s13, offs, tok, R)
    assert r is not None, "the fp8 x fp8 kernel must cover the Air prefill shape"
    act_q, act_bs = r
    y = ops.grouped_gemm_down_fp8a8(act_q, act_bs, q2, s2, offs, scale, E)
    assert y is not None
    torch.cuda.synchronize()
    assert torch.isfinite(y.float()).all()
    offs_c, scale_c = offs.cpu(), scale.cpu()
    nblk = I // 32
    worst = {}
    for e in (0, 31, 63, 64):
        lo, hi = int(offs_c[e]), int(offs_c[e + 1])
        n = hi - lo
        if n == 0:
            continue
        rr = (torch.cat([torch.arange(0, min(256, n)), torch.arange(max(0, n - 150), n)]).unique() + lo)
        rg = rr.cuda()
        src = tok[rg].long()
        qe13 = q13[e].cpu().contiguous().view(torch.float8_e4m3fn)
        qe2 = q2[e].cpu().contiguous().view(torch.float8_e4m3fn)
        xq_c, xs_c = xq[src].cpu().view(torch.float8_e4m3fn), xs[src].cpu()
        a_ref = O.gateup_silu_fp8a8(xq_c, xs_c, qe13, s13[e].cpu())              # bf16, before the MX quantisation
        got_q = act_q[rg].cpu().view(torch.float8_e4m3fn)
        got_sb = O.mx_scale_unrecord(act_bs[rg].cpu(), nblk)
        a_got = O.dequant_mxfp8(got_q, got_sb)
        blockmax = a_ref.float().view(-1, nblk, 32).abs().amax(-1, keepdim=True).expand(-1, -1, 32).reshape(a_ref.shape)
        err = (a_got - a_ref.float()).abs()
        # RNE to e4m3: half a step = 2^-4 of the element's binade (normals), 2^-10 of the scaled block for subnormals (the
        # scaled block maximum is >= 224); + three bf16 ulps for the two evaluations' fp32 summation orders
        # ... and the absolute slack of the weight-only test (atol 2^-7: where the gate is far below zero silu is ill-conditioned
        # in ulp terms and the two evaluations' gate values differ by an ulp)
        lim = (2 ** -4 + 2 ** -6) * a_ref.float().abs() + 2 ** -15 * blockmax + 2 ** -7
        bad = err > lim
        assert not bool(bad.any()), (f"fp8a8 T={T} act group {e}: {int(bad.sum())}/{bad.numel()} outside one e4m3 step; worst err/lim "
                                     f"{float((err / lim).max()):.3g} at ref {float(a_ref.float().flatten()[(err / lim).argmax()]):.4g}, "
                                     f"rows {sorted(set(torch.where(bad)[0].tolist()))[:8]} cols {sorted(set(torch.where(bad)[1].tolist()))[:8]}")
        ref_q, ref_sb = O.quantize_mxfp8(a_ref)
        sb_equal = float((got_sb == ref_sb).float().mean())
        q_equal = float((got_q.view(torch.uint8) == ref_q.view(torch.uint8)).float().mean())
        # the kernel's and the CPU's fp32 summation orders differ: a bf16 act value may move by an ulp, its byte with it
        assert sb_equal >= 0.99 and q_equal >= 0.97, (sb_equal, q_equal)
        y_ref = O.down_fp8a8(got_q, got_sb, qe2, s2[e].cpu())
        # the block-scaled MFMA's dot product on these bytes agrees with float64 arithmetic to bf16 rounding (tools/fp8a8_outlier.py,
        # profiles/r05/fp8a8_outlier.txt: 97.5 % of the unscaled results equal bf16(float64), the rest are one ulp away, and the
        # scaled rows equal bf16(unscaled * routing weight) bit for bit).  One flip of the intermediate bf16(acc * scale_row)
        # can move bf16(that * routing weight) by TWO ulps: max 2 ulp here where the unscaled weight-only test has 1
        mag = torch.matmul(O.dequant_mxfp8(got_q, got_sb).abs(), qe2.to(torch.float32).abs().t()) * s2[e].cpu().unsqueeze(0)
        if e < E:
            y_ref = y_ref * scale_c[rr].unsqueeze(-1)
            mag = mag * scale_c[rr].float().unsqueeze(-1)
        assert_close_bf16(y[rg].cpu(), y_ref, 2 if e < E else 1, 0.06, atol=2 ** -9, what=f"fp8a8 T={T} y group {e}")
        yerr = (y[rg].cpu().float() - y_ref.float()).abs()
        # distance to the weight-only oracle (bf16 activations) on the same rows
        a_w = O.silu_and_mul(O.linear_fp8w(x[src].cpu(), qe13, s13[e].cpu()))
        y_w = O.linear_fp8w(a_w, qe2, s2[e].cpu())
        if e < E:
            y_w = y_w * scale_c[rr].unsqueeze(-1)
        rel = lambda a, b: float((a.float() - b.float()).pow(2).mean().sqrt() / b.float().pow(2).mean().sqrt())
        worst[f"group{e}"] = {"act_vs_fp8a8_oracle": F.err_stats(a_got.to(BF), a_ref), "y_vs_fp8a8_oracle": F.err_stats(y[rg].cpu(), y_ref),
                              "act_bytes_equal": q_equal, "scale_bytes_equal": sb_equal,
                              "y_worst_err_over_sum_of_term_magnitudes": float((yerr / mag.clamp_min(1e-30)).max()),
                              "rms_rel_y_w8a8_vs_weight_only_oracle": rel(y[rg].cpu(), y_w),
                              "rms_rel_act_w8a8_vs_weight_only_oracle": rel(a_got, a_w)}
        assert worst[f"group{e}"]["rms_rel_y_w8a8_vs_weight_only_oracle"] < 0.06
    F.report(f"C5.fp8a8_expert_gemms_T{T}", rows=R, parity="unpinned (no fp8 path in the reference)", worst=worst)


def test_c5_moe_block_w8a8_through_the_module(ops):
    """The opt-in W8A8 path through the product's own module (`config.expert_weight_dtype = expert_activation_dtype =
    "fp8_e4m3"`, models/glm4_moe/moe.py): a prefill-sized MoE block at Air dims takes the fp8 x fp8 kernels, a decode-sized
    step keeps the weight-only ones (the launcher refuses < 192 rows per group and the module never asks below that), and
    the prefill result stays within the activation-quantisation distance of the weight-only result on the same routing
    (rms 2-3 %; bound 6 %).  Parity of the path itself: test_c5_fp8a8_expert_gemms_full_size."""
    from nanovllm.models.glm4_moe.moe import Glm4MoeMoE
    cfg = F.air_cfg(expert_weight_dtype="fp8_e4m3")
    old = torch.get_default_dtype()
    torch.set_default_dtype(BF)
    try:
        with torch.device("cuda"):
            blk = Glm4MoeMoE(cfg)
    finally:
        torch.set_default_dtype(old)
    F.init_seeded(blk, 3)
    blk.quantize_experts_fp8()
    seen = []
    orig = ops.grouped_gemm_gateup_silu_fp8a8

    def spy(*a, **kw):
        r = orig(*a, **kw)
        seen.append(r is not None)
        return r
    ops.grouped_gemm_gateup_silu_fp8a8 = spy
    try:
        x = F.randn_bf16(4096, cfg.hidden_size, seed=77)
        with torch.inference_mode():
            y_w = blk(x)
            assert seen == [], "weight-only configuration must not touch the fp8 x fp8 kernels"
            blk.expert_activation_dtype = "fp8_e4m3"
            y_a = blk(x)
            assert seen == [True], "a prefill-sized step must take the fp8 x fp8 kernels"
            y_d = blk(x[:32])
            assert seen == [True], "a decode-sized step keeps the weight-only kernels"
            blk.expert_activation_dtype = None
            y_dw = blk(x[:32])
        torch.cuda.synchronize()
        assert torch.equal(y_d, y_dw)
        rel = float((y_a.float() - y_w.float()).norm() / y_w.float().norm())
        assert torch.isfinite(y_a.float()).all() and rel < 0.06, rel
        F.report("C5.moe_block_w8a8_vs_weight_only", tokens=4096, rms_rel=rel)
        # the launcher's own refusal (nothing enqueued): 32 tokens over 65 groups
        xq, xs = ops.quantize_fp8_rows(x[:32])
        idx, w = blk.gate(x[:32])
        offs, tok, scale, pos = ops.moe_align(idx, w, cfg.n_routed_experts, True)
        w13_q, w13_s, w2_q, w2_s = blk._fp8
        assert orig(xq, xs, w13_q, w13_s, offs, tok, 32 * 9) is None
    finally:
        ops.grouped_gemm_gateup_silu_fp8a8 = orig


def test_c3_paged_attention_full_size(ops):
    """(iv) paged decode attention at B=32, nh=96, nkv=8, ctx=4096 (the no-split instantiation bench.py times) and at
    B=1 ctx=1024 (split + reduce), stand-alone and fused-from-QKV."""
    nh, nkv, hd, rd = 96, 8, 128, 64
    scale = hd ** -0.5
    for tag, B, ctx in (("C3", 32, 4096), ("C2", 1, 1024)):
        caches, bt, cl, pos, slots = F.paged_decode_inputs(B, ctx, nkv, hd, seed=17)
        kc, vc = caches[0]
        q = F.randn_bf16(B, nh, hd, seed=18)
        out = ops.paged_attn_decode(q, kc, vc, cl.cuda(), bt.cuda(), scale)
        torch.cuda.synchronize()
        ref = O.attn_decode_paged(q.cpu(), kc.cpu(), vc.cpu(), cl, bt, scale)
        ref32 = O.attn_decode_paged(q.cpu().float(), kc.cpu().float(), vc.cpu().float(), cl, bt, scale)
        st, st32, so = F.err_stats(out, ref), F.err_stats(out, ref32), F.err_stats(ref, ref32)
        F.report(f"{tag}.paged_attn_decode", B=B, ctx=ctx, vs_oracle=st, vs_fp32=st32, oracle_vs_fp32=so)
        # outputs are averages of ~ctx N(0,1) values (|o| ~ 0.02-0.1): the reference's attention tolerance
        # rtol = atol = 1e-3 (tests/attention.py:37, fp16) is met in absolute terms
        F.assert_allclose_ref_tol(out, ref, f"{tag} paged attention", rtol=1.6e-2, atol=1e-3)
        # fused: rotary + cache store + attention straight from the packed QKV rows == the unfused sequence
        qkv = F.randn_bf16(B, (nh + 2 * nkv) * hd, seed=19, scale=1.3)
        cs = O.build_cos_sin_cache(rd, 4096, 1e6).cuda()
        kc0, vc0, kc1, vc1 = kc.clone(), vc.clone(), kc.clone(), vc.clone()
        qr, _, _ = ops.qkv_split_norm_rope_store(qkv, nh, nkv, hd, pos.cuda(), cs, rd, None, None, 1e-5, kc0, vc0,
                                                 slots.cuda(), False)
        o0 = ops.paged_attn_decode(qr, kc0, vc0, cl.cuda(), bt.cuda(), scale)
        o1 = ops.paged_attn_decode_qkv(qkv, nh, nkv, hd, pos.cuda(), cs, rd, slots.cuda(), kc1, vc1, cl.cuda(),
                                       bt.cuda(), scale)
        assert torch.equal(kc0, kc1) and torch.equal(vc0, vc1) and torch.equal(o0, o1), f"{tag}: fused != unfused"
        # and against the oracle chain split -> rope -> store -> attention
        qc, kcpu, vcpu = qkv.cpu().split([nh * hd, nkv * hd, nkv * hd], dim=-1)
        qo, ko = O.rope_neox(pos, qc.reshape(B, nh, hd), kcpu.reshape(B, nkv, hd), cs.cpu(), rd)
        kr, vr = kc.cpu(), vc.cpu()
        O.store_kvcache(ko, vcpu.reshape(B, nkv, hd), kr, vr, slots)
        assert torch.equal(kc1.cpu(), kr) and torch.equal(vc1.cpu(), vr), "rotary + cache store must be bit-exact"
        ref1 = O.attn_decode_paged(qo, kr, vr, cl, bt, scale)
        F.report(f"{tag}.paged_attn_decode_qkv", vs_oracle=F.err_stats(o1, ref1))
        F.assert_allclose_ref_tol(o1, ref1, f"{tag} fused paged attention", rtol=1.6e-2, atol=1e-3)


# ------------------------------------------------------------------------------------------------------------------
def test_c3_model_logits(ops):
    """Numeric logits parity (north star: "logits within 1e-3 bf16 rtol"): the model wrapper at Air dims, L=2 (layer
    0 dense, layer 1 MoE — the reference's own instantiation, model.py:21-22), decode B=32 ctx=4096, final norm quirk
    and lm_head (model.py:141-145, embed_head.py:57-67).  bf16 logits cannot agree to 1e-3 elementwise (one bf16 ulp
    is 3.9e-3 relative); what is asserted is (a) max |logit error| relative to max |logit| <= 1e-2, (b) the HIP
    logits are no further from the fp32 evaluation than the reference's own bf16 evaluation is, (c) greedy tokens
    agree wherever the oracle's top-2 margin exceeds the observed error.  The 1e-3 figure is printed beside them."""
    from nanovllm.utils.context import reset_context, set_context
    B, ctx = 32, 4096
    cfg = F.air_cfg()
    model = F.build_model(cfg, seed=3)
    model.model.embed_tokens.weight.data.normal_(0.0, 1.0, generator=torch.Generator(device="cuda").manual_seed(4))
    nkv, hd = cfg.num_key_value_heads, cfg.head_dim
    caches, bt, cl, pos, slots = F.paged_decode_inputs(B, ctx, nkv, hd, seed=27, n_layers=2)
    cpu_caches = [(k.cpu(), v.cpu()) for k, v in caches]
    for layer, (kc, vc) in zip(model.model.layers, caches):
        layer.self_attn.attn.k_cache, layer.self_attn.attn.v_cache = kc, vc
    ids = torch.randint(0, cfg.vocab_size, (B,), generator=torch.Generator().manual_seed(28))
    caps = [{} for _ in model.model.layers]
    for layer, cap in zip(model.model.layers, caps):
        if hasattr(layer.mlp, "gate"):
            layer.mlp.register_forward_pre_hook(
                lambda m, a, kw, cap=cap: cap.update(idx=kw["routing"][0]), with_kwargs=True)
    set_context(False, slot_mapping=slots.cuda(), context_lens=cl.cuda(), block_tables=bt.cuda())
    with torch.inference_mode():
        hid = model(ids.cuda(), pos.cuda())
        logits = model.compute_logits(hid)
    reset_context()
    torch.cuda.synchronize()
    from oracle import from_product
    lws = [from_product.layer_weights_from_module(l) for l in model.model.layers]
    cfgd, cs = from_product.cfg_dict(cfg), from_product.cos_sin_cache(cfg)
    meta = O.AttnMeta(False, slot_mapping=slots, context_lens=cl, block_tables=bt)
    embed, normw, headw = (model.model.embed_tokens.weight.detach().cpu(), model.model.norm.weight.detach().cpu(),
                           model.lm_head.weight.detach().cpu())
    traces = [{} for _ in lws]
    href = O.model_forward(ids, pos, embed, lws, normw, cfgd, cs, [(k.clone(), v.clone()) for k, v in cpu_caches],
                           meta, traces=traces)
    lref = O.linear(href, headw)
    forced = [t.get("idx") for t in traces]
    h32 = O.model_forward(ids, pos, embed.float(), [l.to(torch.float32) for l in lws], normw.float(), cfgd, cs,
                          [(k.float(), v.float()) for k, v in cpu_caches], meta, force_idx=forced)
    l32 = O.linear(h32, headw.float())
    # rows where the MoE layer routed like the oracle (others: near-ties, checked by the layer tests)
    keep = [r for r in range(B) if set(caps[1]["idx"][r].tolist()) == set(traces[1]["idx"][r].tolist())]
    rows = torch.tensor(keep, dtype=torch.long)
    st, st32, so = F.err_stats(logits, lref, rows), F.err_stats(logits, l32, rows), F.err_stats(lref, l32, rows)
    lg, lr = logits.float().cpu()[rows], lref.float()[rows]
    top2 = torch.topk(lr, 2, dim=-1)
    margin = top2.values[:, 0] - top2.values[:, 1]
    agree = lg.argmax(-1) == top2.indices[:, 0]
    err_row = (lg - lr).abs().amax(-1)
    F.report("C3.model_logits_L2", rows_compared=len(keep), vs_oracle=st, vs_fp32=st32, oracle_vs_fp32=so,
             north_star_rtol=1e-3, greedy_agree=int(agree.sum()), greedy_rows=len(keep),
             min_margin_where_disagree=float(margin[~agree].min()) if bool((~agree).any()) else None)
    assert len(keep) >= 24
    # cap 6x: a logit of magnitude 4..8 that lands two bf16 ulp (0.0625) from the reference's is within 0.06 + 0.06|ref|
    F.assert_close_like_reference(logits, lref, l32, "logits", rows, cap_factor=6.0)
    assert bool((agree | (margin <= 2 * err_row)).all()), "greedy token differs beyond the logit error"
