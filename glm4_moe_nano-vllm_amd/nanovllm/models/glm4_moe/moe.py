"""Sigmoid top-k router and the MoE block — reference: models/glm4_moe/moe.py:13-176.

The reference loops over all experts in Python with three host syncs per expert
(moe.py:135-151) and re-reads a debug file from disk on every call (:153-157).  Here one
forward is five HIP launches with no host sync (graph-capturable):
  glm_router_topk → glm_moe_align → glm_grouped_gemm_gateup_silu → glm_grouped_gemm_down →
  glm_moe_combine,
with the shared expert riding as group E of the grouped GEMMs.  Expert weights live in two
stacked buffers ([E+1, 2I, H] and [E+1, H, I]); `experts[i].gate_up_proj.weight` etc. are views
into them, so the reference's attribute/state-dict surface is unchanged."""
from __future__ import annotations

import torch
import torch.nn as nn

from nanovllm import hip_ops
from nanovllm.distributed import expert_parallel as EP
from nanovllm.distributed.parallel_state import get_tp_group, unsharded
from nanovllm.utils.loader import load_one
from .mlp import Glm4MoeMLP


class GlmMoeSelectTopk(nn.Module):

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.top_k = config.num_experts_per_tok
        self.n_routed_experts = config.n_routed_experts
        assert config.n_group == 1
        assert config.topk_group == 1
        self.routed_scaling_factor = config.routed_scaling_factor
        self.weight = nn.Parameter(torch.empty((self.n_routed_experts, config.hidden_size)))
        self.register_buffer("e_score_correction_bias",
                             torch.zeros((self.n_routed_experts), dtype=torch.float32))

    def forward(self, hidden_states: torch.Tensor, precomputed=None):
        if precomputed is not None:  # (topk_idx, topk_w) from the fused add-RMSNorm + router launch
            return precomputed
        h = hidden_states.view(-1, self.config.hidden_size)
        # indices come back in ascending expert id per row (torch.topk(sorted=False) leaves
        # the order unspecified; moe.py:39)
        return hip_ops.router_topk(h, self.weight.data, self.e_score_correction_bias, self.top_k,
                                   float(self.routed_scaling_factor))


class Glm4MoeMoE(nn.Module):

    def __init__(self, config):
        super().__init__()
        self.config = config
        mk = lambda: Glm4MoeMLP(hidden_size=config.hidden_size,
                                intermediate_size=config.moe_intermediate_size,
                                hidden_act=config.hidden_act, reduce_results=False)
        # N3 expert parallelism (`config.expert_parallel_size` = TP world size, BASELINE config 4): this rank builds
        # only ITS experts, with their full intermediate size, plus a replica of the shared expert; `experts[i]` is
        # then the expert with global id expert_ids[i] (distributed/expert_parallel.py)
        self.ep_size = int(getattr(config, "expert_parallel_size", 1) or 1)
        E = config.n_routed_experts
        if self.ep_size > 1:
            tp = get_tp_group()
            assert self.ep_size == tp.world_size, "expert_parallel_size must equal the tensor-parallel world size"
            assert E % self.ep_size == 0, "n_routed_experts must divide over the expert-parallel ranks"
            self.expert_ids = list(range(tp.rank * (E // self.ep_size), (tp.rank + 1) * (E // self.ep_size)))
            with unsharded():
                self.experts = nn.ModuleList([mk() for _ in self.expert_ids])
                self.shared_experts = mk()
        else:
            self.expert_ids = list(range(E))
            self.experts = nn.ModuleList([mk() for _ in range(E)])
            self.shared_experts = mk()
        self.gate = GlmMoeSelectTopk(config)
        self._w13 = None
        self._w2 = None
        self._stack(copy=False)
        # N4: `config.expert_weight_dtype = "fp8_e4m3"` streams the experts as OCP fp8 (weight-only,
        # per-row scale); quantised lazily from the loaded bf16 weights at the first forward
        self.expert_weight_dtype = getattr(config, "expert_weight_dtype", None)
        # r05, OPT-IN (default off; the weight-only path stays the parity path): `config.expert_activation_dtype = "fp8_e4m3"`
        # with fp8 expert weights runs prefill-sized steps as W8A8 on the block-scaled fp8 MFMA — x quantised per token, act
        # per (row, 32 columns) by the gate|up epilogue (hip_ops.grouped_gemm_*_fp8a8); decode-sized steps are HBM-bound
        # and keep the weight-only kernels
        self.expert_activation_dtype = getattr(config, "expert_activation_dtype", None)
        self._fp8 = None
        self._bf16_freed = False

    # ---- stacked expert storage ---------------------------------------------------------
    def _mlps(self):
        return list(self.experts) + [self.shared_experts]

    def _stack(self, copy: bool) -> None:
        mlps = self._mlps()
        w13_0, w2_0 = mlps[0].gate_up_proj.weight, mlps[0].down_proj.weight
        if w13_0.numel():  # (empty after the fp8 release: the shapes of the first stacking are kept for a reload)
            self._shapes = (tuple(w13_0.shape), tuple(w2_0.shape))
        s13, s2 = self._shapes
        w13 = hip_ops.padded_weight(*s13, len(mlps), dtype=w13_0.dtype, device=w13_0.device)
        w2 = hip_ops.padded_weight(*s2, len(mlps), dtype=w2_0.dtype, device=w2_0.device)
        for i, m in enumerate(mlps):
            if copy:
                w13[i].copy_(m.gate_up_proj.weight.data)
                w2[i].copy_(m.down_proj.weight.data)
            m.gate_up_proj.weight.data = w13[i]
            m.down_proj.weight.data = w2[i]
        self._w13, self._w2 = w13, w2

    def _stacked(self):
        if self._bf16_freed:
            raise RuntimeError("the bf16 expert stacks were released after fp8 quantisation "
                               "(config.keep_bf16_experts = True keeps them)")
        first = self.experts[0].gate_up_proj.weight
        if self._w13 is None or first.data_ptr() != self._w13.data_ptr():
            self._stack(copy=True)  # parameters were moved/replaced (e.g. .to(device)): re-tie
        return self._w13, self._w2

    def quantize_experts_fp8(self, free_bf16: bool | None = None) -> None:
        """Build the fp8 copies of the stacked expert weights from the current bf16 values.  Unless
        `config.keep_bf16_experts` is set, the bf16 stacks are then RELEASED (the per-expert parameters become empty
        tensors): fp8 experts are there to halve the expert memory, not to add 50 % to it.  After the release a second
        call has nothing to quantise from and is a no-op (the fp8 copies stay); `load_weights` re-allocates the stacks,
        loads and quantises again."""
        if self._bf16_freed and self._fp8 is not None:
            return
        w13, w2 = self._stacked()
        w13_q, w13_s = hip_ops.quantize_fp8_rowwise(w13)
        w2_q, w2_s = hip_ops.quantize_fp8_rowwise(w2)
        self._fp8 = (w13_q, w13_s, w2_q, w2_s)
        if free_bf16 is None:
            free_bf16 = not bool(getattr(self.config, "keep_bf16_experts", False))
        if free_bf16:
            for m in self._mlps():
                m.gate_up_proj.weight.data = w13.new_empty(0)
                m.down_proj.weight.data = w2.new_empty(0)
            self._w13 = self._w2 = None
            self._bf16_freed = True

    # ---- weights ------------------------------------------------------------------------
    def load_weights(self, path: str, prefix: str):
        gw = load_one(path, f"{prefix}.gate.weight")
        if gw is None:
            raise ValueError(f"No tensor '{prefix}.gate.weight' found in '{path}'")
        self.gate.weight.data.copy_(gw)
        gb = load_one(path, f"{prefix}.gate.e_score_correction_bias")
        if gb is not None:
            self.gate.e_score_correction_bias.copy_(gb)
        if self._bf16_freed:  # a reload after the fp8 release: fresh bf16 stacks, re-tied to the per-expert parameters
            self._bf16_freed, self._fp8 = False, None
            self._stack(copy=False)
        self._stacked()
        for expert, gid in zip(self.experts, self.expert_ids):
            expert.load_weights(path, f"{prefix}.experts.{gid}")
        self.shared_experts.load_weights(path, f"{prefix}.shared_experts")
        if self.expert_weight_dtype == "fp8_e4m3":  # per layer, right after loading: the bf16 peak is ONE layer's stack
            self.quantize_experts_fp8()

    def _shared_fp8(self, h: torch.Tensor) -> torch.Tensor:
        """The (replicated) shared expert from its fp8 rows of the stack — one group, no gather, unscaled."""
        w13_q, w13_s, w2_q, w2_s = self._fp8
        El = len(self.experts)
        act = hip_ops.grouped_gemm_gateup_silu_fp8w(h, w13_q[El:El + 1], w13_s[El:El + 1], None, None, h.shape[0])
        return hip_ops.grouped_gemm_down_fp8w(act, w2_q[El:El + 1], w2_s[El:El + 1], None, None, 0)

    def _ep_experts(self, ops=hip_ops, rank: int | None = None):
        El = len(self.experts)
        r = get_tp_group().rank if rank is None else rank
        E, k = self.config.n_routed_experts, self.config.num_experts_per_tok
        if self.expert_weight_dtype == "fp8_e4m3":  # r04: this rank's experts (and the shared one) as fp8 e4m3 weight-only
            if self._fp8 is None:
                self.quantize_experts_fp8()
            w13_q, w13_s, w2_q, w2_s = self._fp8
            return EP.EPExperts(ops, self.ep_size, r, E, k, None, None, self._shared_fp8,
                                fp8=(w13_q[:El], w13_s[:El], w2_q[:El], w2_s[:El]))
        w13, w2 = self._stacked()
        return EP.EPExperts(ops, self.ep_size, r, E, k, w13[:El], w2[:El], self.shared_experts)

    # ---- forward ------------------------------------------------------------------------
    def forward(self, hidden_states: torch.Tensor, routing=None, defer_combine: bool = False, aligned=None, expert_rows=None):
        """routing: (topk_idx, topk_w) already computed for these rows (decode_layer's fused norm + router).
        aligned: hip_ops.moe_align(topk_idx, topk_w, E, True) of that routing, when the same launch produced it.
        defer_combine: the caller hands the result straight to the next layer's RMSNorm.forward(x, residual) and
        accepts a hip_ops.DeferredCombine (TP=1 only).
        expert_rows: y [T k + T, H] of this routing, when the caller's launch has already run the grouped projections
        (one token: hip_ops.decode_mlp_engine); only the combine is left."""
        shape = hidden_states.shape
        h = hidden_states.view(-1, shape[-1])
        T = h.shape[0]
        E, k = self.config.n_routed_experts, self.config.num_experts_per_tok
        topk_idx, topk_w = self.gate(h, precomputed=routing)
        if self.ep_size > 1:
            # `aligned` (a bucketing of ALL tokens over ALL experts) is of no use here: every rank buckets its own slice
            return EP.run(self._ep_experts().forward(h, topk_idx, topk_w), group=get_tp_group().process_group).view(shape)
        offs, tok, scale, pos = aligned if aligned is not None else hip_ops.moe_align(topk_idx, topk_w, E, True)
        R = T * k + T
        if expert_rows is not None:
            assert self.ep_size == 1 and aligned is not None and expert_rows.shape[0] == R
            y = expert_rows
        elif self.expert_weight_dtype == "fp8_e4m3":
            if self._fp8 is None:
                self.quantize_experts_fp8()
            w13_q, w13_s, w2_q, w2_s = self._fp8
            y = None
            if self.expert_activation_dtype == "fp8_e4m3" and R // (E + 1) >= 192:
                xq, xs = hip_ops.quantize_fp8_rows(h)
                r = hip_ops.grouped_gemm_gateup_silu_fp8a8(xq, xs, w13_q, w13_s, offs, tok, R)
                if r is not None:
                    y = hip_ops.grouped_gemm_down_fp8a8(r[0], r[1], w2_q, w2_s, offs, scale, E)
            if y is None:
                act = hip_ops.grouped_gemm_gateup_silu_fp8w(h, w13_q, w13_s, offs, tok, R)
                y = hip_ops.grouped_gemm_down_fp8w(act, w2_q, w2_s, offs, scale, E)
        else:
            w13, w2 = self._stacked()
            # a token's k experts are distinct and the shared group holds T rows: no group exceeds T rows (T = 1 selects
            # the single-sequence GEMV body)
            act = hip_ops.grouped_gemm_gateup_silu(h, w13, offs, tok, R, max_rows_per_group=T)
            y = hip_ops.grouped_gemm_down(act, w2, offs, scale, E, max_rows_per_group=T)
        if defer_combine and get_tp_group().world_size == 1 and shape[-1] % 8 == 0 and shape[-1] <= 16384:
            return hip_ops.DeferredCombine(y, pos, T, k, T * k, shape)
        out = hip_ops.moe_combine(y, pos, T, k, T * k)
        if get_tp_group().world_size > 1:
            get_tp_group().all_reduce(out)
        return out.view(shape)
