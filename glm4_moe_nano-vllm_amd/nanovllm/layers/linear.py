"""Linear layers with the reference's sharding API — reference: layers/linear.py:14-155.
Weights are [out, in] bf16; forward is glm_gemm_bf16 (weight-streaming MFMA kernel for decode
batches, LDS-tiled MFMA kernel for prefill).  TP size/rank come from the real group
(distributed/parallel_state.py); RowParallel all-reduces over RCCL when tp > 1."""
from __future__ import annotations

import torch
from torch import nn

from nanovllm import hip_ops
from nanovllm.distributed.parallel_state import get_tp_group


def divide(numerator, denominator):
    assert numerator % denominator == 0
    return numerator // denominator


class LinearBase(nn.Module):

    def __init__(self, input_size: int, output_size: int, bias: bool = False,
                 tp_dim: int | None = None):
        super().__init__()
        self.tp_dim = tp_dim
        self.tp_rank = get_tp_group().rank
        self.tp_size = get_tp_group().world_size
        # [out, in] view of row-padded storage (hip_ops.padded_weight): same API, better HBM spread
        self.weight = nn.Parameter(hip_ops.padded_weight(output_size, input_size))
        self.weight.weight_loader = self.weight_loader
        if bias:
            self.bias = nn.Parameter(torch.empty(output_size))
            self.bias.weight_loader = self.weight_loader
        else:
            self.register_parameter("bias", None)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError

    def _gemm(self, x: torch.Tensor, with_bias: bool = True) -> torch.Tensor:
        b = self.bias.data if (with_bias and self.bias is not None) else None
        return hip_ops.gemm_bf16(x, self.weight.data, b)


class ReplicatedLinear(LinearBase):

    def __init__(self, input_size: int, output_size: int, bias: bool = False):
        super().__init__(input_size, output_size, bias)

    def weight_loader(self, param: nn.Parameter, loaded_weight: torch.Tensor):
        param.data.copy_(loaded_weight)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._gemm(x)


class ColumnParallelLinear(LinearBase):

    def __init__(self, input_size: int, output_size: int, bias: bool = False):
        tp_size = get_tp_group().world_size
        super().__init__(input_size, divide(output_size, tp_size), bias, 0)

    def weight_loader(self, param: nn.Parameter, loaded_weight: torch.Tensor):
        shard = param.data.size(self.tp_dim)
        param.data.copy_(loaded_weight.narrow(self.tp_dim, self.tp_rank * shard, shard))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._gemm(x)

    # decode batches: the output as fp32 K-slice partial sums WITHOUT the bias, for a consumer kernel that reduces them
    # on load (the fused decode attention).  Measured r02 (Air dims): the QKV GEMM alone 27.2 -> 22.3 us at M=32 and
    # 21.1 -> 19.9 us at M=1 as 2 slices; the consumer's wider loads give part of it back: layer 474.5 -> 472.2-473.5 us
    # at bs=32, 141.9 -> 143.3 us at bs=1 — hence only from 17 rows on.
    partial_slices = 2
    partial_min_rows = 17

    def can_defer_reduce(self, x: torch.Tensor) -> bool:
        return (x.dim() == 2 and self.partial_min_rows <= x.shape[0] <= 32 and x.shape[1] >= 2048
                and x.shape[1] % 8 == 0 and x.is_contiguous())

    def forward_partial(self, x: torch.Tensor):
        return hip_ops.gemm_bf16_partial(x, self.weight.data, self.partial_slices)


class MergedColumnParallelLinear(ColumnParallelLinear):

    def __init__(self, input_size: int, output_sizes: list[int], bias: bool = False):
        self.output_sizes = output_sizes
        super().__init__(input_size, sum(output_sizes), bias)

    def weight_loader(self, param: nn.Parameter, loaded_weight: torch.Tensor, loaded_shard_id: int):
        offset = sum(self.output_sizes[:loaded_shard_id]) // self.tp_size
        size = self.output_sizes[loaded_shard_id] // self.tp_size
        dst = param.data.narrow(self.tp_dim, offset, size)
        dst.copy_(loaded_weight.chunk(self.tp_size, self.tp_dim)[self.tp_rank])


class QKVParallelLinear(ColumnParallelLinear):
    """q heads are split over the ranks; kv heads too while there are at least as many kv heads as ranks, otherwise each
    kv head is REPLICATED on tp_size / total_num_kv_heads consecutive ranks (attention_new.py:21-34 sizes the attention
    that way; the reference's own loader never gets there because its TP group is a stub)."""

    def __init__(self, hidden_size: int, head_size: int, total_num_heads: int,
                 total_num_kv_heads: int | None = None, bias: bool = False):
        tp_size = get_tp_group().world_size
        total_num_kv_heads = total_num_kv_heads or total_num_heads
        self.head_size = head_size
        self.num_heads = divide(total_num_heads, tp_size)
        if total_num_kv_heads >= tp_size:
            self.num_kv_heads, self.kv_replicas = divide(total_num_kv_heads, tp_size), 1
        else:
            self.num_kv_heads, self.kv_replicas = 1, divide(tp_size, total_num_kv_heads)
        # per-rank rows (not total / tp_size once kv heads are replicated)
        LinearBase.__init__(self, hidden_size, (self.num_heads + 2 * self.num_kv_heads) * head_size, bias, 0)

    def weight_loader(self, param: nn.Parameter, loaded_weight: torch.Tensor, loaded_shard_id: str):
        assert loaded_shard_id in ["q", "k", "v"]
        q_rows = self.num_heads * self.head_size
        kv_rows = self.num_kv_heads * self.head_size
        offset, size = {"q": (0, q_rows), "k": (q_rows, kv_rows), "v": (q_rows + kv_rows, kv_rows)}[loaded_shard_id]
        dst = param.data.narrow(self.tp_dim, offset, size)
        if loaded_shard_id == "q":
            dst.copy_(loaded_weight.chunk(self.tp_size, self.tp_dim)[self.tp_rank])
        else:
            shards = self.tp_size // self.kv_replicas
            dst.copy_(loaded_weight.chunk(shards, self.tp_dim)[self.tp_rank // self.kv_replicas])


class RowParallelLinear(LinearBase):

    def __init__(self, input_size: int, output_size: int, bias: bool = False):
        tp_size = get_tp_group().world_size
        super().__init__(divide(input_size, tp_size), output_size, bias, 1)

    def weight_loader(self, param: nn.Parameter, loaded_weight: torch.Tensor):
        shard = param.data.size(self.tp_dim)
        param.data.copy_(loaded_weight.narrow(self.tp_dim, self.tp_rank * shard, shard))

    # K-slice partial sums pay off when few rows meet a long reduction (o_proj at decode: 32 x 12288 -> 4096
    # runs 21 us instead of 31 us); the caller must hand the result straight to RMSNorm.forward(x, residual)
    partial_min_k = 8192
    partial_slices = 8

    @staticmethod
    def _partial_rows(m: int) -> bool:
        lo = 1 if hip_ops.opt("tinym") != 0 else 0  # one row: the GEMV body (glm_gemm_bf16 picks it)
        return lo < m <= 32

    def forward(self, x: torch.Tensor, defer_reduce: bool = False):
        # (not for a single row: the single-sequence GEMV body streams the whole reduction per row and needs no K split)
        if (defer_reduce and self.tp_size == 1 and self.bias is None and x.dim() == 2 and self._partial_rows(x.shape[0])
                and x.shape[1] >= self.partial_min_k and x.shape[1] >= 2 * self.weight.shape[0]
                and x.shape[1] % 8 == 0 and x.stride(1) == 1):
            return hip_ops.gemm_bf16_partial(x, self.weight.data, self.partial_slices)
        y = self._gemm(x, with_bias=self.tp_rank == 0)
        if self.tp_size > 1:
            get_tp_group().all_reduce(y)
        return y
