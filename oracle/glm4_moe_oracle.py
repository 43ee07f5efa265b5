"""CPU oracle for the GLM-4-MoE per-layer forward path of 2419322417/GLM4_MOE_nano-vllm.

TEST INFRASTRUCTURE ONLY.  This file restates, with plain torch-CPU tensor ops, the algorithm
of the reference's hot path with the reference's exact rounding points, each function citing
the reference file:line it follows (paths relative to /root/reference/nanovllm/).  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it,
and only as the checker — the product (glm4_moe_nano-vllm_amd/) never does and has no CPU path.

Pinning: the reference ships no golden vectors.  The oracle is pinned by
``tests/golden/*.safetensors``, minted by importing the reference itself in the build container
(``tests/golden/gen_golden.py``); ``tests/test_oracle_vs_golden.py`` asserts bit equality for
every function below.  Arithmetic living in third-party code absent from the reference tree
(flash-attn, pyproject.toml:18, unpinned; call sites layers/attention.py:97-107) is PARITY
UNPINNED: the goldens use an fp32 softmax(QK^T)V stand-in, restated in ``attn_*`` here.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------
# a2 / a3  layers/layernorm.py:17-40
def rms_forward(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """layernorm.py:17-26 — fp32 statistics, cast back, THEN multiply by the weight."""
    dt = x.dtype
    xf = x.to(torch.float32)
    inv = torch.rsqrt(xf.square().mean(dim=-1, keepdim=True) + eps)
    return (xf * inv).to(dt) * weight


def add_rms_forward(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float):
    """layernorm.py:29-40 — the variance is taken from the UNROUNDED fp32 sum."""
    dt = x.dtype
    s = x.to(torch.float32) + residual.to(torch.float32)
    new_residual = s.to(dt)
    inv = torch.rsqrt(s.square().mean(dim=-1, keepdim=True) + eps)
    return (s * inv).to(dt) * weight, new_residual


# a4  layers/RMSNorm.py:19-33
def head_rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """RMSNorm.py:29-33 — weight multiply in fp32 BEFORE the down-cast (differs from a2)."""
    dt = x.dtype
    xf = x.to(torch.float32)
    xf = xf * torch.rsqrt(xf.square().mean(dim=-1, keepdim=True) + eps)
    return (weight * xf).to(dt)


# ---------------------------------------------------------------------------------------------
# a6  layers/rotary_embedding.py
def build_cos_sin_cache(rotary_dim: int, max_position: int, base: float) -> torch.Tensor:
    """rotary_embedding.py:30-36 — fp32 [max_pos, 1, rotary_dim] = cos | sin."""
    expo = torch.arange(0, rotary_dim, 2, dtype=torch.float) / rotary_dim
    inv_freq = 1.0 / (base ** expo)
    t = torch.arange(max_position, dtype=torch.float)
    fr = torch.outer(t, inv_freq)
    return torch.cat((fr.cos(), fr.sin()), dim=-1).unsqueeze(1)


def _rotate_neox(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """rotary_embedding.py:6-14 — half-split rotation in fp32."""
    a, b = x.to(torch.float32).chunk(2, dim=-1)
    return torch.cat((a * cos - b * sin, b * cos + a * sin), dim=-1).to(x.dtype)


def rope_neox(positions, q, k, cos_sin_cache, rotary_dim: int):
    """rotary_embedding.py:39-55 — rotate the first rotary_dim dims of each head, keep the rest."""
    cs = cos_sin_cache[positions]  # [T,1,rd]
    cos, sin = cs.chunk(2, dim=-1)
    hd = q.shape[-1]
    outs = []
    for t in (q, k):
        rot, keep = t.split([rotary_dim, hd - rotary_dim], dim=-1)
        outs.append(torch.cat((_rotate_neox(rot, cos, sin), keep), dim=-1))
    return outs[0], outs[1]


def rotary_dim_of(head_size: int, partial_rotary_factor: float) -> int:
    """rotary_embedding.py:68-69."""
    return int(head_size * partial_rotary_factor) if partial_rotary_factor < 1.0 else head_size


# ---------------------------------------------------------------------------------------------
# a7  layers/attention.py:10-42
def store_kvcache(key, value, k_cache, v_cache, slot_mapping) -> None:
    """attention.py:20-30 — row scatter by slot, slot -1 skipped (in place)."""
    N = key.shape[0]
    D = key.shape[1] * key.shape[2]
    kc, vc = k_cache.view(-1, D), v_cache.view(-1, D)
    for i in range(N):
        s = int(slot_mapping[i])
        if s == -1:
            continue
        kc[s] = key[i].reshape(D)
        vc[s] = value[i].reshape(D)


def _attend(q, k, v, scale: float, causal_offset: int | None):
    """fp32 softmax(scale·QK^T)·V for one sequence.  q [Tq,nh,hd], k/v [Tk,nkv,hd].
    causal_offset = Tk - Tq (bottom-right aligned mask) or None for no mask."""
    nh, nkv = q.shape[1], k.shape[1]
    rep = nh // nkv
    qf = q.to(torch.float32).transpose(0, 1)                            # [nh,Tq,hd]
    kf = k.to(torch.float32).transpose(0, 1).repeat_interleave(rep, 0)  # [nh,Tk,hd]
    vf = v.to(torch.float32).transpose(0, 1).repeat_interleave(rep, 0)
    s = torch.matmul(qf, kf.transpose(1, 2)) * scale
    dead = None
    if causal_offset is not None:
        Tq, Tk = q.shape[0], k.shape[0]
        i = torch.arange(Tq).unsqueeze(1)
        j = torch.arange(Tk).unsqueeze(0)
        s = s.masked_fill(j > i + causal_offset, float("-inf"))
        dead = (j > i + causal_offset).all(dim=-1) if Tk > 0 else torch.ones(Tq, dtype=torch.bool)
    p = torch.softmax(s, dim=-1)
    if dead is not None and bool(dead.any()):
        # more queries than keys: flash-attn's published rule for the bottom-right aligned mask (its README, "2.1: change
        # behavior of causal flag") — a row whose mask is all zero gives an all-zero output row, not the NaN of a softmax
        # over nothing.  (The reference never issues such a call; the kernels are tested on it.)
        p = p.masked_fill(dead.view(1, -1, 1), 0.0)
    return torch.matmul(p, vf).transpose(0, 1).to(q.dtype)              # [Tq,nh,hd]


def _gather_paged(cache, block_table_row, length: int):
    """cache [nblk, bs, nkv, hd] → the first `length` tokens of one sequence."""
    bs = cache.shape[1]
    nb = (length + bs - 1) // bs
    blocks = [cache[int(block_table_row[j])] for j in range(nb)]
    return torch.cat(blocks, dim=0)[:length]


def attn_decode_paged(q, k_cache, v_cache, context_lens, block_tables, scale: float):
    """a8  attention.py:101-107 (flash_attn_with_kvcache; stand-in, parity unpinned).
    q [B,nh,hd] → [B,nh,hd]; rows with context_len 0 give zeros."""
    out = torch.zeros_like(q)
    for b in range(q.shape[0]):
        n = int(context_lens[b])
        if n == 0:
            continue
        k = _gather_paged(k_cache, block_tables[b], n)
        v = _gather_paged(v_cache, block_tables[b], n)
        out[b] = _attend(q[b:b + 1], k, v, scale, None)[0]
    return out


def attn_prefill_varlen(q, k, v, cu_seqlens_q, cu_seqlens_k, scale: float,
                        block_tables=None):
    """a9  attention.py:85-100 (flash_attn_varlen_func, causal; stand-in, parity unpinned).
    With block_tables, k/v are the paged caches (prefix-cache hit, attention.py:86-87)."""
    out = torch.empty_like(q)
    B = cu_seqlens_q.numel() - 1
    for b in range(B):
        q0, q1 = int(cu_seqlens_q[b]), int(cu_seqlens_q[b + 1])
        k0, k1 = int(cu_seqlens_k[b]), int(cu_seqlens_k[b + 1])
        if block_tables is None:
            kk, vv = k[k0:k1], v[k0:k1]
        else:
            kk = _gather_paged(k, block_tables[b], k1 - k0)
            vv = _gather_paged(v, block_tables[b], k1 - k0)
        out[q0:q1] = _attend(q[q0:q1], kk, vv, scale, (k1 - k0) - (q1 - q0))
    return out


# ---------------------------------------------------------------------------------------------
# a5 / a10 / a13  layers/linear.py, layers/activation.py, models/glm4_moe/mlp.py
def linear(x, weight, bias=None):
    """linear.py:53,75,152 — F.linear in the tensor dtype (fp32 accumulate, one rounding)."""
    return F.linear(x, weight, bias)


def silu_and_mul(x):
    """activation.py:12-14 — silu(gate) is rounded to the tensor dtype before the multiply."""
    g, u = x.chunk(2, dim=-1)
    return F.silu(g) * u


def mlp(x, w_gate_up, w_down):
    """mlp.py:49-54 — gate_up (rows gate|up, mlp.py:118) → SiluAndMul → down."""
    return linear(silu_and_mul(linear(x, w_gate_up)), w_down)


# ---------------------------------------------------------------------------------------------
# a11  models/glm4_moe/moe.py:28-50
def router(x, gate_w, e_score_correction_bias, top_k: int, routed_scaling_factor: float, force_idx=None):
    """R1..R7 of SURVEY.md §8a: bf16 logits → bf16 sigmoid → +fp32 bias → top-k (unsorted) →
    gather un-biased scores → renormalise (always) → scale.  `force_idx` (test aid, no reference counterpart)
    replaces the top-k DECISION R4 only; the weights R5-R7 are still computed here from this call's scores."""
    h = x.reshape(-1, x.shape[-1])
    logits = F.linear(h, gate_w)
    scores = logits.sigmoid()
    choice = scores + e_score_correction_bias.unsqueeze(0)
    idx = torch.topk(choice, k=top_k, dim=-1, sorted=False)[1] if force_idx is None else force_idx
    w = scores.gather(1, idx)
    w = w / (w.sum(dim=-1, keepdim=True) + 1e-20)
    w = w * routed_scaling_factor
    return idx, w


def router_scores(x, gate_w, e_score_correction_bias):
    """The intermediate tensors of `router` (logits, scores, choice) for tie/ambiguity analysis."""
    h = x.reshape(-1, x.shape[-1])
    logits = F.linear(h, gate_w)
    scores = logits.sigmoid()
    return logits, scores, scores + e_score_correction_bias.unsqueeze(0)


# a12  models/glm4_moe/moe.py:128-176
def moe(x, gate_w, bias, top_k, routed_scaling_factor, w13, w2, shared_w13, shared_w2,
        topk=None, force_idx=None, trace=None):
    """Per-expert loop in ascending expert id, accumulation in the tensor dtype (index_add_),
    then + shared expert.  w13 [E,2I,H], w2 [E,H,I].  `topk` overrides the routing decision
    (used to isolate the expert arithmetic from router rounding in tests)."""
    h = x.reshape(-1, x.shape[-1])
    idx, w = router(h, gate_w, bias, top_k, routed_scaling_factor, force_idx) if topk is None else topk
    if trace is not None:
        trace.update(h_router=h, idx=idx, w=w)
    acc = torch.zeros_like(h)
    for e in range(w13.shape[0]):
        hit = idx == e
        if not bool(hit.any()):
            continue
        rows, slot = torch.where(hit)
        y = mlp(h[rows], w13[e], w2[e])
        acc.index_add_(0, rows, y * w[rows, slot].unsqueeze(-1))
    return acc + mlp(h, shared_w13, shared_w2)


# ---------------------------------------------------------------------------------------------
# a1 + attention wrapper  models/glm4_moe/attention_new.py:281-364, decode_layer.py:79-100
class LayerWeights:
    """Plain container for one decoder layer's tensors (names follow the reference modules)."""

    def __init__(self, **kw):
        self.input_ln = kw["input_ln"]
        self.post_ln = kw["post_ln"]
        self.qkv_w = kw["qkv_w"]
        self.qkv_b = kw.get("qkv_b")
        self.o_w = kw["o_w"]
        self.q_norm = kw.get("q_norm")
        self.k_norm = kw.get("k_norm")
        # MoE layer
        self.gate_w = kw.get("gate_w")
        self.gate_bias = kw.get("gate_bias")
        self.w13 = kw.get("w13")
        self.w2 = kw.get("w2")
        self.shared_w13 = kw.get("shared_w13")
        self.shared_w2 = kw.get("shared_w2")
        # dense layer
        self.dense_w13 = kw.get("dense_w13")
        self.dense_w2 = kw.get("dense_w2")
        # N4: fp8 weight-only experts (q13 [E+1,2I,H] float8_e4m3fn, s13 [E+1,2I] fp32, q2 [E+1,H,I], s2 [E+1,H]; the
        # shared expert is entry E); when set, the routed and shared experts use these instead of w13/w2
        self.fp8 = kw.get("fp8")


    def to(self, dtype):
        """Same tensors in another floating dtype (integer/None fields untouched).  Every function of this file is
        dtype-generic: with fp32 tensors each `.to(dt)` / F.linear rounding point of the reference vanishes, so
        `lw.to(torch.float32)` + fp32 activations IS the rounding-free fp32 evaluation of the same algorithm
        (used by the tests to state the HIP path's error against exact arithmetic next to its error against the
        reference's bf16 arithmetic)."""
        kw = {k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v)
              for k, v in self.__dict__.items()}
        if kw.get("fp8") is not None:  # the quantised bytes stay; only the scales follow the dtype
            q13, s13, q2, s2 = kw["fp8"]
            kw["fp8"] = (q13, s13.to(dtype) if dtype == torch.float32 else s13, q2, s2.to(dtype) if dtype == torch.float32 else s2)
        return LayerWeights(**kw)


class AttnMeta:
    """The fields of utils/context.py:5-14 the attention reads."""

    def __init__(self, is_prefill, slot_mapping=None, context_lens=None, block_tables=None,
                 cu_seqlens_q=None, cu_seqlens_k=None):
        self.is_prefill = is_prefill
        self.slot_mapping = slot_mapping
        self.context_lens = context_lens
        self.block_tables = block_tables
        self.cu_seqlens_q = cu_seqlens_q
        self.cu_seqlens_k = cu_seqlens_k


def attention_block(h, positions, lw: LayerWeights, cfg, cos_sin_cache, k_cache, v_cache,
                    meta: AttnMeta):
    """attention_new.py:326-364 + layers/attention.py:64-109."""
    nh, nkv, hd = cfg["num_attention_heads"], cfg["num_key_value_heads"], cfg["head_dim"]
    qkv = linear(h, lw.qkv_w, lw.qkv_b)
    q, k, v = qkv.split([nh * hd, nkv * hd, nkv * hd], dim=-1)
    q = q.reshape(-1, nh, hd)
    k = k.reshape(-1, nkv, hd)
    v = v.reshape(-1, nkv, hd)
    if lw.q_norm is not None:
        q = head_rmsnorm(q, lw.q_norm, cfg["rms_norm_eps"])
        k = head_rmsnorm(k, lw.k_norm, cfg["rms_norm_eps"])
    rd = rotary_dim_of(hd, cfg.get("partial_rotary_factor", 0.5))
    q, k = rope_neox(positions, q, k, cos_sin_cache, rd)
    scale = hd ** -0.5
    if k_cache is not None and k_cache.numel():
        store_kvcache(k, v, k_cache, v_cache, meta.slot_mapping)
    if meta.is_prefill:
        if meta.block_tables is not None:
            o = attn_prefill_varlen(q, k_cache, v_cache, meta.cu_seqlens_q, meta.cu_seqlens_k,
                                    scale, meta.block_tables)
        else:
            o = attn_prefill_varlen(q, k, v, meta.cu_seqlens_q, meta.cu_seqlens_k, scale)
    else:
        o = attn_decode_paged(q, k_cache, v_cache, meta.context_lens, meta.block_tables, scale)
    return linear(o.flatten(1, -1), lw.o_w)


def decoder_layer(positions, h, residual, lw: LayerWeights, cfg, cos_sin_cache, k_cache,
                  v_cache, meta: AttnMeta, force_idx=None, trace=None, fp32_inplace_alias: bool = False):
    """decode_layer.py:79-100 — returns (mlp_out, residual); mlp_out is NOT added to the
    residual here (the next layer's add_rms does it).  `trace` (a dict) receives the intermediates the tests
    compare (attention output, router input, routing); `force_idx` is handed to `router`."""
    eps = cfg["rms_norm_eps"]
    # layernorm.py:21-25,35-39 work IN PLACE on `x.float()`.  In the runner's bf16 regime (model_runner.py:30) that is a
    # copy; with fp32 tensors it is the caller's tensor itself, so the reference AS EXECUTED in fp32 returns a residual
    # that aliases the normalised output (`residual = x.to(orig_dtype)` is x, then `x.mul_` scales both).  That is an
    # artefact of fp32 module dtype, not the algorithm: only `fp32_inplace_alias=True` (the C1 fixture's fp32 leg)
    # reproduces it; the default stays the dtype-generic evaluation the tests use as exact arithmetic.
    alias = fp32_inplace_alias and h.dtype == torch.float32
    if residual is None:
        h, residual = rms_forward(h, lw.input_ln, eps), h
        if alias:
            residual = h
    else:
        h, residual = add_rms_forward(h, residual, lw.input_ln, eps)
        if alias:
            residual = h
    h = attention_block(h, positions, lw, cfg, cos_sin_cache, k_cache, v_cache, meta)
    if trace is not None:
        trace["attn_out"] = h
    h, residual = add_rms_forward(h, residual, lw.post_ln, eps)
    if alias:
        residual = h
    if lw.gate_w is not None and getattr(lw, "fp8", None) is not None:
        # N4: same router, experts evaluated on the dequantised fp8 weights (moe_fp8w)
        hh = h.reshape(-1, h.shape[-1])
        idx, w = router(hh, lw.gate_w, lw.gate_bias, cfg["num_experts_per_tok"], cfg["routed_scaling_factor"], force_idx)
        if trace is not None:
            trace.update(h_router=hh, idx=idx, w=w)
        q13, s13, q2, s2 = lw.fp8
        E = q13.shape[0] - 1
        h = moe_fp8w(hh, cfg["num_experts_per_tok"], idx, w, q13[:E], s13[:E], q2[:E], s2[:E], q13[E], s13[E], q2[E],
                     s2[E]).reshape(h.shape)
    elif lw.gate_w is not None:
        h = moe(h, lw.gate_w, lw.gate_bias, cfg["num_experts_per_tok"],
                cfg["routed_scaling_factor"], lw.w13, lw.w2, lw.shared_w13, lw.shared_w2,
                force_idx=force_idx, trace=trace)
    else:
        h = mlp(h, lw.dense_w13, lw.dense_w2)
    return h, residual


def model_forward(input_ids, positions, embed_w, layers, final_norm_w, cfg, cos_sin_cache,
                  kv_caches, meta: AttnMeta, final_norm_without_residual: bool = True, force_idx=None,
                  traces=None):
    """models/glm4_moe/model.py:64-78 — note the final norm is applied to the last MLP output
    WITHOUT the residual (model.py:75-76, the add+norm line is commented out there);
    `final_norm_without_residual=False` evaluates that commented-out add+norm instead (what a full-depth
    checkpoint needs).  force_idx / traces: per-layer lists handed to decoder_layer."""
    h = F.embedding(input_ids, embed_w)
    residual = None
    for i, lw in enumerate(layers):
        kc, vc = kv_caches[i] if kv_caches is not None else (None, None)
        h, residual = decoder_layer(positions, h, residual, lw, cfg, cos_sin_cache, kc, vc, meta,
                                    force_idx=None if force_idx is None else force_idx[i],
                                    trace=None if traces is None else traces[i])
    if final_norm_without_residual or residual is None:
        return rms_forward(h, final_norm_w, cfg["rms_norm_eps"])
    return add_rms_forward(h, residual, final_norm_w, cfg["rms_norm_eps"])[0]


def greedy_sample(logits):
    """layers/sampler.py:13-16 (all temperatures == 0)."""
    return logits.to(torch.float32).argmax(dim=-1)


# ---------------------------------------------------------------------------------------------
# N4  fp8 (OCP e4m3fn) weight-only quantised experts.  No reference counterpart (the reference has
# int4-AWQ dequant-then-GEMM, layers/linear_awq_new.py:254-255; BASELINE config 5 asks for fp8):
# the oracle is the reference's expert arithmetic (a12/a13) evaluated on the DEQUANTISED weights.
FP8_MAX = 448.0


def quantize_fp8_rowwise(w: torch.Tensor):
    """Per-row scale s = max|row|/448, q = e4m3fn(row/s) (round-to-nearest-even, saturating)."""
    s = (w.abs().amax(dim=-1).to(torch.float32) / FP8_MAX).clamp_min(1e-12)
    q = (w.to(torch.float32) / s.unsqueeze(-1)).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn)
    return q, s


def linear_fp8w(x, q, s):
    """fp32 accumulation of bf16(x) * fp8(q), row scale on the fp32 sum, ONE rounding to x.dtype."""
    acc = torch.matmul(x.to(torch.float32), q.to(torch.float32).t())
    return (acc * s.unsqueeze(0)).to(x.dtype)


def mlp_fp8w(x, q13, s13, q2, s2):
    return linear_fp8w(silu_and_mul(linear_fp8w(x, q13, s13)), q2, s2)


def moe_fp8w(x, top_k, idx, w, q13, s13, q2, s2, sq13, ss13, sq2, ss2):
    """moe() with quantised experts; routing (idx, w) is given (the router stays bf16)."""
    h = x.reshape(-1, x.shape[-1])
    acc = torch.zeros_like(h)
    for e in range(q13.shape[0]):
        hit = idx == e
        if not bool(hit.any()):
            continue
        rows, slot = torch.where(hit)
        y = mlp_fp8w(h[rows], q13[e], s13[e], q2[e], s2[e])
        acc.index_add_(0, rows, y * w[rows, slot].unsqueeze(-1))
    return acc + mlp_fp8w(h, sq13, ss13, sq2, ss2)


# ---------------------------------------------------------------------------------------------
# N4 on the fp8 MFMA (r05): W8A8 — fp8 e4m3 activations next to the fp8 weights, block-scaled MFMA.  PARITY UNPINNED by
# construction: the reference has no fp8 path at all (its quantised linears are int4 AWQ dequant-then-GEMM,
# layers/linear_awq_new.py:254-255).  The oracle is the reference's expert arithmetic (mlp.py:49-54, moe.py:149) evaluated
# on the DEQUANTISED weights AND the DEQUANTISED activations, with the quantisers restated here (OCP MX for the act rows);
# tests/test_hip_kernels.py reports the distance of this leg to the bf16-activation oracle (linear_fp8w) next to it.
def quantize_fp8_rows(x: torch.Tensor):
    """glm_quantize_fp8_rows: per-row scale s = max(amax / 448, 2^-126) in fp32, q = e4m3fn(RNE(clamp(x / s)))."""
    xf = x.to(torch.float32)
    s = (xf.abs().amax(dim=-1) / FP8_MAX).clamp_min(1.17549435e-38)
    q = (xf / s.unsqueeze(-1)).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn)
    return q, s


def quantize_mxfp8(v: torch.Tensor):
    """The gate|up epilogue's MX quantiser: per (row, 32 columns) an E8M0 scale byte sb with
    2^(sb - 127) = 2^(floor(log2 amax) - 8) (OCP MX, e4m3 emax = 8) or the next power of two where amax would pass 448
    (nothing saturates); sb = 127 for an all-zero or subnormal block, clamped to [1, 253]; q = e4m3fn(RNE(v * 2^(127 - sb))).  Returns (q [R, I] float8, sb [R, I / 32] uint8)."""
    R, I = v.shape
    assert I % 32 == 0
    vf = v.to(torch.float32).view(R, I // 32, 32)
    amax = vf.abs().amax(dim=-1)
    bits = amax.view(torch.int32)
    eb = (bits >> 23) & 0xff
    bump = ((bits & 0x7fffff) > 0x600000).to(torch.int32)   # amax / 2^(e - 8) > 448: one step up so that nothing saturates
    sb = torch.where(eb == 0, torch.full_like(eb, 127), (eb - 8 + bump).clamp(1, 253))
    inv = ((254 - sb) << 23).to(torch.int32).view(torch.float32)
    q = (vf * inv.unsqueeze(-1)).clamp(-FP8_MAX, FP8_MAX).to(torch.float8_e4m3fn).view(R, I)
    return q, sb.to(torch.uint8)


def mx_scale_record(sb: torch.Tensor) -> torch.Tensor:
    """[R, nblk] scale bytes -> the [R, 64] record of the C-ABI: block b at (b & 3) * 16 + (b >> 2)."""
    R, nb = sb.shape
    rec = torch.zeros(R, 64, dtype=torch.uint8)
    b = torch.arange(nb)
    rec[:, (b & 3) * 16 + (b >> 2)] = sb
    return rec


def mx_scale_unrecord(rec: torch.Tensor, nblk: int) -> torch.Tensor:
    b = torch.arange(nblk)
    return rec[:, (b & 3) * 16 + (b >> 2)]


def dequant_mxfp8(q: torch.Tensor, sb: torch.Tensor) -> torch.Tensor:
    R, I = q.shape
    sc = torch.exp2(sb.to(torch.float32) - 127.0)
    return (q.to(torch.float32).view(R, I // 32, 32) * sc.unsqueeze(-1)).view(R, I)


def gateup_silu_fp8a8(xq, xs, q13, s13):
    """act = bf16(bf16(silu(bf16 gate)) * bf16 up) (mlp.py:49-53 via activation.py:13-15) with
    gate|up = (fp32 sum of fp8 x fp8 products) * weight-row scale * token scale, one rounding to bf16."""
    acc = torch.matmul(xq.to(torch.float32), q13.to(torch.float32).t())
    gu = (acc * s13.unsqueeze(0) * xs.unsqueeze(-1)).to(torch.bfloat16)
    return silu_and_mul(gu)


def down_fp8a8(act_q, act_sb, q2, s2):
    """y = bf16((fp32 sum over k of w2_q * act_q * 2^(sb - 127)) * weight-row scale) — mlp.py:53 on the dequantised operands."""
    acc = torch.matmul(dequant_mxfp8(act_q, act_sb), q2.to(torch.float32).t())
    return (acc * s2.unsqueeze(0)).to(torch.bfloat16)
